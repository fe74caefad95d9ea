"""Copy the summaries of gpurun_out/prof/<tag>/ (tools/profile_round.sh) into profiles/<dest>/ and regenerate
profiles/traffic.json, which bench.py reads for roofline.traffic / roofline.issue while the kernel sources are unchanged.

    python tools/install_profiles.py r02_b_pmc c4_ram c4_ram_target c2_dram ...
"""
import hashlib, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


sys.path.insert(0, ROOT)
from mcmcf90_amd.build import source_sha as sha  # noqa: E402


dest, tags = sys.argv[1], sys.argv[2:]
dst = os.path.join(ROOT, "profiles", dest)
os.makedirs(dst, exist_ok=True)
tfile = os.path.join(ROOT, "profiles", "traffic.json")
tj = {}
for t in tags:
    src = os.path.join(ROOT, "gpurun_out", "prof", t)
    j = json.load(open(os.path.join(src, "summary.json")))
    if j["kernels_sha"] != sha():
        print("skipping %s: profiled with other kernel sources (%s, now %s)" % (t, j["kernels_sha"], sha()))
        continue
    os.makedirs(os.path.join(dst, t), exist_ok=True)
    for f in ("summary.json", "kernel_stats.csv", "bench_kt.json"):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(dst, t, f))
    e = {k: j[k] for k in ("kernels_sha", "hbm_bytes_per_proposal", "hbm_read_bytes_per_proposal", "hbm_write_bytes_per_proposal",
                           "valu_insts_per_proposal", "salu_insts_per_proposal", "valu_busy", "wait_any") if k in j}
    e["profile"] = "profiles/%s/%s/summary.json" % (dest, t)
    e["source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in separate passes (tools/profile_round.sh), MI355X; read bytes = "
                   "2 x FETCH_SIZE (gfx950 correction for coalesced streams); per proposal = median per launch of the dominant kernel / "
                   "proposals per launch")
    tj[t] = e
json.dump(tj, open(tfile, "w"), indent=1)
print(json.dumps({k: (round(v.get("hbm_bytes_per_proposal", 0)), round(v.get("valu_insts_per_proposal", 0))) for k, v in tj.items()}))
