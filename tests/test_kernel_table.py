"""The engine's kernel-selection tables (mcx_api.hip: STEP_TABLE / GROUP_TABLE / SCAM_TABLE, exported through
mcmcx_debug_kernel_table) against the GPU suite: every selectable sampling-kernel instance must be named in an assertion on
`last_kernel()` by some parity test -- so that a size- or switch-dependent instance cannot ship with no test that runs it
(pooled_mfma_kernel<false, true> did).  The static half runs on CPU (the table needs no device); the dynamic half
(tests/test_zz_kernel_coverage.py) requires that the GPU session really launched each of them."""
import glob
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))


def _table():
    from mcmcf90_amd.engine import kernel_table
    return kernel_table()


def test_table_lists_the_three_launchers():
    tab = _table()
    fams = {f for f, _ in tab}
    assert fams == {"step", "group", "scam"}
    names = [n for _, n in tab]
    assert len(names) == len(set(names)) >= 25                 # every entry its own name: mcmcx_last_kernel tells instances apart
    for must in ("pooled_mfma_kernel<false, true>", "step_kernel_ram_wide", "group_step_kernel<quad, DR2>", "scam_pooled12_kernel"):
        assert must in names


def test_every_entry_is_asserted_by_a_gpu_parity_test():
    src = {}
    for f in glob.glob(os.path.join(HERE, "test_gpu_*.py")):
        with open(f) as fh:
            text = fh.read()
        if "last_kernel()" in text and "pytest.mark.gpu" in text:
            src[os.path.basename(f)] = text
    missing = []
    for fam, name in _table():
        lit = re.escape('"%s"' % name)
        if not any(re.search(lit, t) for t in src.values()):
            missing.append("%s:%s" % (fam, name))
    assert not missing, "kernel instances no -m gpu test asserts by name: %s" % ", ".join(missing)


def test_tools_and_bench_parse():
    """The measurement scripts under tools/ (and bench.py, __graft_entry__.py) are evidence for DESIGN.md's numbers: they must at least compile."""
    import ast
    import glob
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "tools", "*.py"))) + [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]
    assert len(files) > 20
    for f in files:
        ast.parse(open(f).read(), filename=f)
