#!/bin/bash
# round 6, GPU call c: the suite on the FIFO build, the tick's kernels, the response-column kernel at 1 / 2 / 3 waves per SIMD, the host-callback break-even, the variants
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06c; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > $O/gpu_suite.txt 2>&1; echo "suite rc $?"; tail -n 3 $O/gpu_suite.txt
tools/tick_ab.sh r06c > $O/tick_ab.txt 2>&1; cat $O/tick_ab.txt
for v in "" cols2 cols3; do
  if [ -n "$v" ]; then export MCMCX_LIBRARY=$PWD/variants_build/libmcmcx_$v.so; fi
  for w in c1x c1; do
    python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$w lib=${v:-tree} value %.4g ms/step %.3f kernel %s' % (j['value'], j['ms_per_step'], r['kernel']))"
  done
done > $O/cols_waves.txt 2>&1; unset MCMCX_LIBRARY; cat $O/cols_waves.txt
MCMCX_LIBRARY=$PWD/variants_build/libmcmcx_neg.so python tools/variants/check_variants.py > $O/check_variants.txt 2>&1; echo "variants rc $?"; tail -n 3 $O/check_variants.txt
timeout -k 10 400 python tools/hostcb_breakeven.py > $O/hostcb_breakeven.txt 2>&1; echo "breakeven rc $?"; cat $O/hostcb_breakeven.txt
