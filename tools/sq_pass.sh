#!/bin/bash
# tools/sq_pass.sh TAG [bench args] -- one rocprofv3 --pmc pass of SQ counters (no tracing) over a short bench.py run; per-dispatch sums of the sampling kernel
TAG=$1; shift; REPO=$PWD; OUT=$REPO/gpurun_out/sq/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/a -o pmc -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs "$@" > $OUT/a.json 2> $OUT/a.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/b -o pmc -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs "$@" > $OUT/b.json 2> $OUT/b.err
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/g -o pmc -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs "$@" > $OUT/g.json 2> $OUT/g.err
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, os, sys, statistics, json
out = sys.argv[1]
for p in ("a", "b", "g"):
    acc = {}
    for f in glob.glob(os.path.join(out, p, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r.get("Kernel_Name", "").split("(")[0]
            if "pooled_mfma" not in kn and "step_kernel" not in kn and "group_step" not in kn:
                continue
            key = (kn[-40:], r["Counter_Name"], r["Dispatch_Id"])
            acc[key] = acc.get(key, 0.0) + float(r["Counter_Value"])
    res = {}
    for (kn, c, _), v in acc.items():
        res.setdefault((kn, c), []).append(v)
    for (kn, c), v in sorted(res.items()):
        print("%-42s %-28s median %.4g over %d dispatches" % (kn, c, statistics.median(v), len(v)))
    try:
        j = json.loads(open(os.path.join(out, p + ".json")).read().strip().splitlines()[-1]); print("   bench under this pass: %.4g proposals/s, %.2f ms/step, avg launch %.2f ms" % (j["value"], j["ms_per_step"], j["roofline"]["avg_launch_ms"]))
    except Exception as ex:
        print("   (no bench line: %s)" % ex)
PY
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
