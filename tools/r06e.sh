#!/bin/bash
# round 6, GPU call e: on the reverted covariance kernels -- the adaptation parity subset + the new tests, the response-column kernel at 1 / 2 waves
# per SIMD, the variants (incl. the covariance FIFO) against the library, the host-callback break-even
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06e; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_gpu_group.py tests/test_gpu_parity.py tests/test_gpu_multirank.py -m gpu -x -q --durations=5 > $O/parity_subset.txt 2>&1; echo "parity subset rc $?"; tail -n 9 $O/parity_subset.txt
for rep in 1 2; do for v in "" cols2; do
  if [ -n "$v" ]; then export MCMCX_LIBRARY=$PWD/variants_build/libmcmcx_$v.so; else unset MCMCX_LIBRARY; fi
  for w in c1x c1; do
    python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$w lib=${v:-tree} value %.4g ms/step %.3f kernel %s launch_ms %.3f share %.3f' % (j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['kernel_share_of_wall']))"
  done
done; done > $O/cols_waves.txt 2>&1; unset MCMCX_LIBRARY; cat $O/cols_waves.txt
MCMCX_LIBRARY=$PWD/variants_build/libmcmcx_neg.so python tools/variants/check_variants.py > $O/check_variants.txt 2>&1; echo "variants rc $?"; tail -n 5 $O/check_variants.txt
timeout -k 10 500 python tools/hostcb_breakeven.py > $O/hostcb_breakeven.txt 2>&1; echo "breakeven rc $?"; cat $O/hostcb_breakeven.txt
