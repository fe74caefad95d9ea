"""Runs LAST in the GPU suite (file name): every entry of the engine's kernel-selection tables must have been launched by some test of
this session (tests/conftest.py notes mcmcx_last_kernel after every Engine.run).  Only meaningful for a whole-suite run: skipped when the
session was narrowed (-k, a file list) or thinned (MCMCX_THIN=1)."""
import os
import pytest

pytestmark = pytest.mark.gpu


def test_every_table_entry_was_launched_in_this_session(request):
    import conftest
    from mcmcf90_amd.engine import kernel_table
    cfg = request.config
    whole = not (getattr(cfg.option, "keyword", "") or "") and all(os.path.basename(os.path.normpath(a)) in ("tests", "") or os.path.isdir(a) for a in cfg.args)
    if not whole or os.environ.get("MCMCX_THIN") == "1" or "slow" in (getattr(cfg.option, "markexpr", "") or ""):
        pytest.skip("not a whole-suite run")
    seen = conftest.KERNELS_SEEN
    missing = [n for _, n in kernel_table() if n not in seen]
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "kernels_seen.txt"), "w") as fh:
            for n in sorted(seen):
                fh.write("%s  %d runs\n" % (n, seen[n]))
    except OSError:
        pass
    assert not missing, "selectable kernel instances no test of this session launched: %s" % ", ".join(missing)
