# tools/c2ab.sh -- run on the GPU box: config 2 (Gaussian d=10, AM) with state + factor in LDS, state only, neither
cd "$(dirname "$0")/.." || exit 1
for v in "MCMCX_X=0" "MCMCX_LDS_SCRATCH=1" "MCMCX_LDS_SCRATCH=0"; do
  echo "== $v"
  env $v python bench.py --workload c2 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('value %.4g ms/step %.3f launch_ms %.3f frac %.3f' % (j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac']))"
done
