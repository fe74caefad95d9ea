#!/bin/bash
# per-kernel times of tools/svd_tick_probe.py NCHAINS (rocprofv3 --kernel-trace --stats)
REPO=$PWD; OUT=$REPO/gpurun_out/kstats_$$; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python3 $REPO/tools/svd_tick_probe.py "$@" > $OUT/out.txt 2>&1
cd $REPO
head -1 $OUT/out.txt
python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'mcx::' in r['Name'] and float(r['Percentage']) > 0.3:
            print("  %-26s calls %3s total %9.1f ms avg %8.3f" % (r['Name'].split('(')[0][-26:], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6))
PY
rm -rf $OUT
