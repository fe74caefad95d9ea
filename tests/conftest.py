import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


GPU_SUITE_BUDGET_S = 600.0          # the driver's step limit for `pytest -m gpu` is 900 s (with -x): stay well inside it
_t_session = [0.0]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: the long GPU cases (npar >= 200 SVD adaptations, full-size runs); thinned with MCMCX_THIN=1 "
                                       "or deselected with -m 'gpu and not slow' when the suite nears its wall-clock budget")
    config.addinivalue_line("markers", "extended: duplicates of covered paths at larger sizes / non-production forms (npar 255 / 256 SVD, the lane SVD at "
                                       "npar 200, two of the twelve-wave SCAM layouts): run with MCMCX_EXTENDED=1 only, so that the default GPU suite "
                                       "keeps >= 20 % headroom under its wall-clock budget")
    # torch bundles its own HIP runtime under the same soname as /opt/rocm's: whichever is loaded first serves the
    # whole process.  Let torch initialise first (as bench.py does) so tests may use torch.cuda next to libmcmcx.so.
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except Exception:
        pass


KERNELS_SEEN = {}               # sampling-kernel name (mcmcx_last_kernel) -> number of Engine.run calls of this session that ended on it


def _note_kernels():
    """Every Engine.run of the session notes which table entry it launched (tests/test_zz_kernel_coverage.py)."""
    try:
        from mcmcf90_amd import engine as _eng
    except Exception:
        return
    if getattr(_eng.Engine.run, "_noting", False):
        return
    inner = _eng.Engine.run

    def run(self, *a, **kw):
        r = inner(self, *a, **kw)
        try:
            k = self.last_kernel()
            if k:
                KERNELS_SEEN[k] = KERNELS_SEEN.get(k, 0) + 1
        except Exception:
            pass
        return r
    run._noting = True
    _eng.Engine.run = run


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


def pytest_sessionstart(session):
    import time
    _t_session[0] = time.time()
    _note_kernels()


def pytest_collection_modifyitems(config, items):
    # MCMCX_THIN=1: drop the cases marked slow (the evidence pipeline's emergency brake: a timed-out suite erases everything)
    if os.environ.get("MCMCX_THIN") == "1":
        skip = pytest.mark.skip(reason="MCMCX_THIN=1: slow case thinned out")
        for it in items:
            if "slow" in it.keywords:
                it.add_marker(skip)
    if os.environ.get("MCMCX_EXTENDED") != "1":
        skip = pytest.mark.skip(reason="extended case: MCMCX_EXTENDED=1 runs it")
        for it in items:
            if "extended" in it.keywords:
                it.add_marker(skip)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Wall-clock budget of the GPU suite: above GPU_SUITE_BUDGET_S the run says so loudly, and with MCMCX_SUITE_BUDGET_STRICT=1
    (tools/final_check.sh, the builder's own pre-flight) it FAILS -- the driver's limit is 900 s and a suite that times out there
    leaves no evidence at all."""
    import time
    dt = time.time() - _t_session[0]
    mexpr = getattr(config.option, "markexpr", "") or ""
    if "gpu" in mexpr and "not gpu" not in mexpr:
        slow = sorted(((r.duration, r.nodeid) for r in terminalreporter.stats.get("passed", []) if getattr(r, "when", "") == "call"), reverse=True)[:8]
        terminalreporter.write_line("GPU suite wall clock %.0f s of a %.0f s budget; slowest: %s" % (dt, GPU_SUITE_BUDGET_S, ", ".join("%s %.0fs" % (n.split("::")[-1][:48], d) for d, n in slow)))
        if dt > GPU_SUITE_BUDGET_S:
            terminalreporter.write_line("GPU SUITE OVER BUDGET: %.0f s > %.0f s -- thin the cases marked `slow` (MCMCX_THIN=1) or split them" % (dt, GPU_SUITE_BUDGET_S), red=True)
            if os.environ.get("MCMCX_SUITE_BUDGET_STRICT") == "1":
                terminalreporter._session.exitstatus = 1


# Which sampling-kernel family a GPU test runs.  The engine picks the lane-group kernels (mcx_group.hpp) by itself wherever they cover the
# configuration and win -- which is everywhere at the chain counts of a test.  The modules below predate them and exercise the
# lane-per-chain kernels (their LDS / global-scratch / delayed-rejection variants, selected through other switches), so they run with
# MCMCX_GROUP=0 unless a test asks for "auto" through the `kernels` parameter; tests/test_gpu_group.py compares the two families
# directly, and the Fortran-shim, run1 and multi-rank modules take whatever the engine picks.
# (pooled mode, host callbacks, user modules and SCAM are outside the lane-group kernels' coverage: those modules run the engine's own choice)
_LANE_MODULES = ("test_gpu_parity", "test_gpu_primitives", "test_gpu_fullsize", "test_gpu_fuzz")


@pytest.fixture(autouse=True)
def _kernel_family(request, monkeypatch):
    mod = request.module.__name__.split(".")[-1]
    fam = None
    if hasattr(request.node, "callspec"):
        fam = request.node.callspec.params.get("kernels")
    if fam == "auto":
        monkeypatch.delenv("MCMCX_GROUP", raising=False)
    elif fam == "lane" or mod in _LANE_MODULES:
        monkeypatch.setenv("MCMCX_GROUP", "0")
    yield
