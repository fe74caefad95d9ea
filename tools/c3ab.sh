# tools/c3ab.sh -- run on the GPU box: config 3 (banana d=20, DRAM) with the delayed-rejection kernel, build variants of it, and the general step kernel
cd "$(dirname "$0")/.." || exit 1
for v in "MCMCX_X=0" "MCMCX_LIBRARY=$PWD/tools/_build/libmcmcx_q2nb2.so" "MCMCX_LIBRARY=$PWD/tools/_build/libmcmcx_q2nb1.so" "MCMCX_DR_GENERAL=1"; do
  echo "== $v"
  env $v python bench.py --workload c3 --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('value %.4g ms/step %.3f launch_ms %.3f frac %.3f' % (j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac']))"
done
