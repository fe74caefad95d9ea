mkdir -p gpurun_out/rw
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/rw/base.json 2> gpurun_out/rw/base.err
for v in a b c; do MCMCX_LIBRARY=$PWD/_variants/lib$v.so python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/rw/$v.json 2> gpurun_out/rw/$v.err; done
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/rw/base2.json 2> gpurun_out/rw/base2.err
for v in base a b c base2; do python - <<PY
import json
j=json.load(open("gpurun_out/rw/$v.json")); print("$v", "%.4g"%j["value"], j["ms_per_step"], j["roofline"]["frac"])
PY
done
