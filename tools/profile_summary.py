"""Summarise one configuration profiled by tools/profile_round.sh (gpurun_out/prof/TAG): kernel-trace stats and the PMC
passes, per launch of the dominant kernel and per proposal.  Prints JSON; with --traffic KEY it also prints the entry
for profiles/traffic.json (bench.py reads it as `roofline.traffic` / `roofline.issue` as long as the kernel sources are
unchanged).

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a coalesced stream,
so the read bytes are 2 x FETCH_SIZE (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for streaming stores."""
import csv, glob, hashlib, json, os, statistics, sys

out = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("step_kernel", "scam_pooled_kernel", "scam_pooled12_kernel", "scam_kernel", "pooled_mfma_kernel", "pooled_mfma_ks_kernel")


sys.path.insert(0, ROOT)
from mcmcf90_amd.build import source_sha as kernels_sha  # noqa: E402


def bench_line(tag):
    try:
        return json.loads(open(os.path.join(out, "bench_%s.json" % tag)).read().strip().splitlines()[-1])
    except Exception as e:
        return {"error": str(e)}


def counters(tag):
    """{counter: median over the dominant kernel's dispatches of the per-dispatch sum}; the dominant kernel = the sampling kernel
    whose dispatches sum to the most over the pass (a launch may queue several instantiations of which all but one return at once)"""
    acc, tot = {}, {}
    for f in glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r.get("Kernel_Name", "")
            if not any(k in kn for k in KERNELS):
                continue
            name = kn.split("(")[0]
            key = (name, r["Counter_Name"], r["Dispatch_Id"])
            acc[key] = acc.get(key, 0.0) + float(r["Counter_Value"])
            tot[name] = tot.get(name, 0.0) + float(r["Counter_Value"])
    name = max(tot, key=tot.get) if tot else None
    res = {}
    for (kn, c, _), v in acc.items():
        if kn == name:
            res.setdefault(c, []).append(v)
    return {c: statistics.median(v) for c, v in res.items()}, {c: len(v) for c, v in res.items()}, name


res = {"kernels_sha": kernels_sha()}
for f in glob.glob(os.path.join(out, "kt", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    res["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "AverageNs", "Percentage")} for r in rows if "mcx::" in r.get("Name", "")][:8]
line = bench_line("kt")
res["bench_line_kt"] = line
per_launch_prop = None
try:
    r = line["roofline"]
    proposals = line["value"] * line["ms_per_step"] / 1e3 * line["steps"]
    per_launch_prop = proposals / line["n_gpus"] / max(r["launches"], 1)
    res["proposals_per_launch"] = per_launch_prop
    res["avg_launch_ms_unprofiled_events"] = r["avg_launch_ms"]
except Exception as e:
    res["error"] = str(e)
for tag in ("fetch", "write", "sq", "grbm"):
    med, n, name = counters(tag)
    res[tag] = {"median_per_launch": med, "dispatches": n, "kernel": name}
try:
    rd = 2.0 * res["fetch"]["median_per_launch"]["FETCH_SIZE"] * 1024.0
    wr = res["write"]["median_per_launch"]["WRITE_SIZE"] * 1024.0
    res["hbm_read_bytes_per_proposal"] = rd / per_launch_prop
    res["hbm_write_bytes_per_proposal"] = wr / per_launch_prop
    res["hbm_bytes_per_proposal"] = (rd + wr) / per_launch_prop
except Exception as e:
    res["hbm_error"] = str(e)
try:
    sq = res["sq"]["median_per_launch"]
    res["valu_insts_per_proposal"] = sq["SQ_INSTS_VALU"] * 64.0 / per_launch_prop     # lane-instructions (64 per wave instruction)
    res["salu_insts_per_proposal"] = sq["SQ_INSTS_SALU"] / per_launch_prop
    res["valu_busy"] = sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"]               # share of wave-cycles with a VALU instruction in flight
    res["wait_any"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
    res["wait_inst_any"] = sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"]
except Exception as e:
    res["sq_error"] = str(e)
if "--traffic" in sys.argv:
    key = sys.argv[sys.argv.index("--traffic") + 1]
    ent = {k: res[k] for k in ("kernels_sha", "hbm_bytes_per_proposal", "hbm_read_bytes_per_proposal", "hbm_write_bytes_per_proposal",
                               "valu_insts_per_proposal", "salu_insts_per_proposal", "valu_busy", "wait_any") if k in res}
    res["traffic_entry"] = {key: ent}
print(json.dumps(res, indent=1))
