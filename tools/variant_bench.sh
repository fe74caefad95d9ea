#!/bin/bash
# run bench.py (target regime and default) for each experimental build given on the command line
for v in "$@"; do
  for st in target default; do
    MCMCX_LIBRARY=$PWD/tools/_build/libmcmcx_$v.so python bench.py --steps 6 --warmup 2 --no-cpu-baseline --start $st 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$v', '$st', '%.3e'%j['value'], 'ms/launch %.2f'%r['avg_launch_ms'])"
  done
done
