#!/bin/bash
# tools/svd_pmc.sh NCHAINS -- kernel trace and SQ counters of one per-chain SVD adaptation at npar = 200 (tools/svd_tick_probe.py)
N=${1:-4096}; REPO=$PWD; OUT=$REPO/gpurun_out/svdpmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $REPO/tools/svd_tick_probe.py $N > $OUT/kt.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq1 -o pmc -- python3 $REPO/tools/svd_tick_probe.py $N > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/sq2 -o pmc -- python3 $REPO/tools/svd_tick_probe.py $N > $OUT/sq2.log 2>&1
cd $REPO
python3 - <<'PY'
import csv, glob, collections, statistics
base = "gpurun_out/svdpmc"
f = glob.glob(base + "/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("   %-60s calls %4s avg %9.3f ms %6s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
for tag in ("sq1", "sq2"):
    fs = glob.glob(base + "/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs: print(tag, "no counters"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "svd_sweep" in k or "svd_applyv" in k:
            acc[("sweep" if "sweep" in k else "applyv", r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for (k, d), c in acc.items():
        for cn, v in c.items(): by[k][cn].append(v)
    for k in by: print(tag, k, {cn: "%.4g" % statistics.median(v) for cn, v in by[k].items()})
PY
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete
