#!/bin/bash
REPO=$PWD; OUT=$REPO/gpurun_out/kstats_svd; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python3 $REPO/tools/svd_tick_probe.py 4096 > $OUT/probe.txt 2>&1
cd $REPO
head -2 $OUT/probe.txt
python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print("  %-34s calls %4s avg %9.3f ms total %9.1f ms" % (r['Name'].split('(')[0][-34:], r['Calls'], float(r['AverageNs'])/1e6, float(r['TotalDurationNs'])/1e6))
PY
