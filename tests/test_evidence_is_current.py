"""The committed evidence describes the committed sources: `profiles/kernel_resources.txt` (written by every build) and `profiles/traffic.json`
(the PMC figures bench.py reports as `roofline.traffic` / `roofline.issue`) carry the engine sha they were made with -- a kernel edit without a
rebuild / re-profile shows up here, on CPU, instead of as a null `traffic` in the driver's bench line."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sha():
    from mcmcf90_amd.build import source_sha
    return source_sha()


def test_kernel_resources_report_is_of_these_sources():
    text = open(os.path.join(ROOT, "profiles", "kernel_resources.txt")).read()
    m = re.search(r"engine sha ([0-9a-f]{16})", text)
    assert m and m.group(1) == _sha(), "profiles/kernel_resources.txt was written by another build: rebuild (mcmcf90_amd.build.build) and commit it"
    rows = [l for l in text.splitlines() if l and not l.startswith("#") and not l.startswith("kernel ")]
    assert len(rows) > 100
    from mcmcf90_amd.engine import kernel_table
    names = " ".join(rows)
    for fam, name in kernel_table():                    # every selectable kernel is a kernel of the report (instantiations share a base name)
        base = re.split(r"[<\[]", name)[0]
        assert base in names, (fam, name)
    # -Werror=pass-failed keeps declared occupancies honest; the report must not show a spilling headline kernel at one wave
    head = [l for l in rows if l.startswith("step_kernel_ram_wide ")]
    assert head and int(head[0].split()[-1]) >= 2


def test_stored_pmc_traffic_is_of_these_sources():
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert {"c4_ram", "c4_pooled", "c2_dram", "c3_dram", "c5_pooled"} <= set(tj)
    for key, e in tj.items():
        assert e["kernels_sha"] == _sha(), "profiles/traffic.json[%s] was measured on other kernel sources: tools/profile_all.sh + tools/install_profiles.py" % key
        assert os.path.exists(os.path.join(ROOT, e["profile"])), e["profile"]
        assert e["hbm_bytes_per_proposal"] > 0 and e["valu_insts_per_proposal"] > 0, key


def test_kernel_sources_stay_within_140_columns():
    """tools/rewrap.py brought mcmcf90_amd/csrc to at most 140 columns (round 6); what is left are printf formats, inline asm / _Pragma lines and the
    two-line banners of the host parts.  No new long lines."""
    import glob
    long_lines = []
    for f in sorted(glob.glob(os.path.join(ROOT, "mcmcf90_amd", "csrc", "*.h*"))):
        for n, line in enumerate(open(f).read().split("\n"), 1):
            if len(line) > 140:
                long_lines.append("%s:%d (%d)" % (os.path.basename(f), n, len(line)))
    assert len(long_lines) <= 21, long_lines
