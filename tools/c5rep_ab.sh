mkdir -p gpurun_out/c5
run(){ python bench.py --workload c5 --replicas --scam-fast --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/c5/$1.json 2> gpurun_out/c5/$1.err; python - $1 <<'PY'
import json, sys
j = json.load(open("gpurun_out/c5/%s.json" % sys.argv[1])); print(sys.argv[1], "%.4g" % j["value"], j["roofline"]["kernel"])
PY
}
for i in 1 2; do run rep12_$i; MCMCX_SCAM_POOLED_16=1 run rep16_$i; done
