#!/bin/bash
# tools/kstats.sh LIBVARIANT|"" bench-args... : per-kernel times (rocprofv3 --kernel-trace --stats) of one bench.py run
V=$1; shift
REPO=$PWD; OUT=$REPO/gpurun_out/kstats_$$; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ -n "$V" ]; then export MCMCX_LIBRARY=$REPO/tools/_build/libmcmcx_$V.so; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python3 $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > /dev/null 2>&1
cd $REPO
python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'mcx::' in r['Name'] and float(r['Percentage']) > 0.5:
            print("  %-30s calls %3s avg %8.3f min %8.3f max %8.3f ms" % (r['Name'].split('(')[0][-30:], r['Calls'], float(r['AverageNs'])/1e6, float(r['MinNs'])/1e6, float(r['MaxNs'])/1e6))
PY
rm -rf $OUT
