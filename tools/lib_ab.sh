#!/bin/bash
# tools/lib_ab.sh "<bench.py args>" ["<more args>" ...] -- each argument string is one configuration: the in-tree library against every
# variants_build/libmcmcx_*.so, alternating, twice, on the SAME box
mkdir -p gpurun_out/lab
run(){ tag=$1; shift; python bench.py --no-cpu-baseline --no-other-configs $@ > gpurun_out/lab/$tag.json 2> gpurun_out/lab/$tag.err; python - $tag <<'PY'
import json, sys
try:
    j = json.load(open("gpurun_out/lab/%s.json" % sys.argv[1])); print("%-28s %.4g proposals/s  %.3f ms/step  share %.3f  %s" % (sys.argv[1], j["value"], j["ms_per_step"], j["roofline"]["kernel_share_of_wall"], j["roofline"]["kernel"]))
except Exception as ex:
    print(sys.argv[1], "FAILED", ex)
PY
}
c=0
for args in "$@"; do
  c=$((c+1))
  echo "== $args"
  for i in 1 2; do
    run cfg${c}_tree$i $args
    for f in variants_build/libmcmcx_*.so; do v=$(basename $f .so); v=${v#libmcmcx_}; MCMCX_LIBRARY=$PWD/$f run cfg${c}_${v}$i $args; done
  done
done
