#!/bin/bash
# tools/c5rep_waves.sh -- per-chain config 5 (the reference's SCAM, 65536 chains x npar 200) with 1 / 2 / 4 / 8 waves per tile
for w in 2 1 4 8 2; do
  MCMCX_SCAM_WAVES=$w timeout -k 10 200 python bench.py --workload c5 --replicas --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('waves $w value %.4g ms/step %.1f frac %.3f %s' % (j['value'], j['ms_per_step'], r['frac'], r['kernel']))"
done
