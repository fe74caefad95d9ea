"""Summarise gpurun_out/prof_* (tools/profile_headline.sh): kernel-trace stats and PMC HBM bytes per launch of the step kernel."""
import csv, glob, json, os, sys

out = sys.argv[1]
res = {}
for f in glob.glob(os.path.join(out, "prof_kt", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    res["kernel_stats"] = [r for r in rows if "mcx::" in r.get("Name", "")][:8]
for tag in ("fetch", "write"):
    per = []
    meta = {}
    for f in glob.glob(os.path.join(out, "prof_" + tag, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if "step_kernel" not in r.get("Kernel_Name", ""):
                continue
            k = r.get("Dispatch_Id")
            acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
            meta = {"vgpr": r.get("VGPR_Count"), "accum_vgpr": r.get("Accum_VGPR_Count"), "scratch": r.get("Scratch_Size"),
                    "lds": r.get("LDS_Block_Size"), "counter": r.get("Counter_Name")}
        per = [acc[k] for k in sorted(acc, key=lambda x: int(x))]
    res[tag] = {"per_launch": per, **meta}
for tag in ("kt", "fetch", "write"):
    try:
        res["bench_" + tag] = json.loads(open(os.path.join(out, "prof_%s_bench.json" % tag)).read().strip().splitlines()[-1])
    except Exception as e:
        res["bench_" + tag] = str(e)
print(json.dumps(res, indent=1))
