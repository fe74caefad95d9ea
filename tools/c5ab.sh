# tools/c5ab.sh -- config 5 pooled (reference order and scam_fast): the in-tree library against every _variants/lib*.so, alternating
mkdir -p gpurun_out/c5
run(){ python bench.py --workload c5 $2 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/c5/$1.json 2> gpurun_out/c5/$1.err; python - $1 <<'PY'
import json, sys
j = json.load(open("gpurun_out/c5/%s.json" % sys.argv[1])); print(sys.argv[1], "%.4g" % j["value"], round(j["roofline"]["frac"], 4))
PY
}
for i in 1 2; do
  run tree_ref$i ""; run tree_fast$i "--scam-fast"
  for f in _variants/lib*.so; do v=$(basename $f .so); MCMCX_LIBRARY=$PWD/$f run ${v}_ref$i ""; MCMCX_LIBRARY=$PWD/$f run ${v}_fast$i "--scam-fast"; done
done
