#!/bin/bash
# tools/pooled_ab.sh [bench args] -- config 4 pooled (or whatever the args select) with the in-tree library and every variants_build/libmcmcx_*.so,
# alternating, twice, on the SAME box: value and ms per step per library
mkdir -p gpurun_out/pab
run(){ python bench.py --pooled --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs "${@:2}" > gpurun_out/pab/$1.json 2> gpurun_out/pab/$1.err; python - $1 <<'PY'
import json, sys
try:
    j = json.load(open("gpurun_out/pab/%s.json" % sys.argv[1])); print("%-24s %.4g proposals/s  %.2f ms/step  %s" % (sys.argv[1], j["value"], j["ms_per_step"], j["roofline"]["kernel"]))
except Exception as ex:
    print(sys.argv[1], "FAILED", ex)
PY
}
for i in 1 2; do
  run tree$i "$@"
  for f in variants_build/libmcmcx_*.so; do v=$(basename $f .so); v=${v#libmcmcx_}; [ "$v" = phase ] && continue; MCMCX_LIBRARY=$PWD/$f run ${v}$i "$@"; done
done
