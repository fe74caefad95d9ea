#!/bin/bash
# tools/tick_ab.sh TAG -- kernel-trace of the three DRAM / AM configurations (per-kernel averages: the adaptation tick's share)
TAG=$1; REPO=$PWD; OUT=$REPO/gpurun_out/tick/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "c4d --method dram" "c3 --workload c3" "c2 --workload c2"; do
  set -- $cfg; n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -o kt -- python3 $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs "$@" > $OUT/$n.json 2> $OUT/$n.err
  f=$(find $OUT/$n -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$OUT/$n.json" $n <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
j = json.load(open(sys.argv[2]))
print(sys.argv[3], "value %.4g ms/step %.3f" % (j["value"], j["ms_per_step"]))
for r in rows[:6]:
    print("   %-70s calls %4s avg %9.3f ms  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
PY
  find $OUT/$n -name "*.csv" -size +1M -delete; find $OUT/$n -name "*.db" -delete
done
