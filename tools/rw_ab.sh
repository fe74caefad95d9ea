# tools/rw_ab.sh ARGS... -- A/B of build variants under _variants/ (MCMCX_LIBRARY) against the in-tree library on one box
mkdir -p gpurun_out/rw
run(){ python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs "${@:2}" > gpurun_out/rw/$1.json 2> gpurun_out/rw/$1.err; python - $1 <<PY
import json,sys
j=json.load(open("gpurun_out/rw/%s.json"%sys.argv[1])); print(sys.argv[1], "%.4g"%j["value"], j["ms_per_step"], j["roofline"]["frac"])
PY
}
run base "$@"
for f in _variants/lib*.so; do v=$(basename $f .so); MCMCX_LIBRARY=$PWD/$f run $v "$@"; done
run base2 "$@"
