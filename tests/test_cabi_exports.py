"""The C-ABI library builds, loads, and exports every symbol include/mcmcx.h declares (CPU-only check)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from mcmcf90_amd import build, _lib
    build.build()
    hdr = open(os.path.join(ROOT, "include", "mcmcx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mcmcx_[A-Za-z0-9_]+)\s*\(", hdr))
    declared -= {"mcmcx_config", "mcmcx_handle", "mcmcx_engine"}
    assert len(declared) >= 30
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIBPATH]).decode()
    exported = set(re.findall(r"\bT (mcmcx_[A-Za-z0-9_]+)", out))
    assert declared <= exported, declared - exported
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = _lib.load()
    assert L.mcmcx_version().startswith(b"mcmcx")


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a HIP device mcmcx_create must fail (no CPU path).  On a GPU box this is skipped."""
    import ctypes as C
    import pytest
    from mcmcf90_amd import _lib, make_config
    L = _lib.load()
    cfg = make_config(2, 1, nsimu=10)
    h = C.c_void_p()
    rc = L.mcmcx_create(C.byref(cfg), C.byref(h))
    if rc == 0:
        L.mcmcx_destroy(h)
        pytest.skip("a GPU is present")
    assert rc < 0 and b"no HIP device" in L.mcmcx_last_error()
