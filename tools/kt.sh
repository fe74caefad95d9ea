#!/bin/bash
# tools/kt.sh TAG script.py [args...] -- rocprofv3 kernel trace of a python script, top kernels with min / avg / max
TAG=$1; shift; REPO=$PWD; OUT=$REPO/gpurun_out/kt/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python3 $REPO/"$@" > $OUT/out.log 2> $OUT/err.log
cd $REPO
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print("   %-74s calls %5s avg %9.3f min %9.3f max %9.3f ms  %6s %%" % (r["Name"][:74], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6, r["Percentage"]))
PY
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
