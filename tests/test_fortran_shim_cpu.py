"""CPU-side checks of the Fortran host shim: it builds with flang, links the C ABI, reads the reference's
namelist with real namelist I/O and the .dat inputs, and -- without a GPU -- stops with the engine's error
text instead of falling back to anything."""
import os
import subprocess
import tempfile
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FDIR = os.path.join(ROOT, "mcmcf90_amd", "fortran")
FLANG = "/opt/rocm/lib/llvm/bin/flang"

NML = """&mcmc
 method = 'dram'
 nsimu = 100
 updatesigma = 0
 verbosity = 1
/
&mcmcx
 devtarget = 'banana'
 nchains = 64
/
"""


@pytest.mark.skipif(not os.path.exists(FLANG), reason="flang not available")
def test_shim_builds_and_reads_inputs():
    from mcmcf90_amd import build
    build.build()
    subprocess.check_call(["make", "-s", "-C", FDIR])
    syms = subprocess.check_output(["nm", os.path.join(FDIR, "libmcmcxf.a")]).decode()
    for sym in ("mcmc_main_", "mcmc_main_one_", "ssfunction_", "checkbounds_", "priorfun_", "mcx_ss_adapter"):     # mcmc_main.F90:12,49
        assert sym in syms, sym
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "mcmcinit.nml"), "w").write(NML)
        open(os.path.join(d, "mcmcpar.dat"), "w").write("# start\n0 0 0, 0\n")
        open(os.path.join(d, "mcmccov.dat"), "w").write("1 0 0 0\n0 1 0 0\n0 0 1 0\n0 0 0 1\n")
        p = subprocess.run([os.path.join(FDIR, "demo_main")], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           timeout=120)
        out = p.stdout.decode(errors="replace")
    assert "MCMC code version" in out and "using nmlfile mcmcinit.nml" in out
    import torch
    if not torch.cuda.is_available():
        assert "no HIP device" in out and p.returncode != 0          # loud failure, no fallback
    else:
        assert p.returncode == 0, out


@pytest.mark.skipif(not os.path.exists(FLANG), reason="flang not available")
def test_shim_rejects_bad_namelist():
    subprocess.check_call(["make", "-s", "-C", FDIR])
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "mcmcinit.nml"), "w").write("&mcmc\n nsimu = 100\n nosuchvariable = 3\n/\n")
        p = subprocess.run([os.path.join(FDIR, "demo_main")], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           timeout=120)
        out = p.stdout.decode(errors="replace")
    assert "Error reading mcmc namelist" in out and "No mcmc run" in out      # mcmcinit.F90:129-137
    assert p.returncode != 0


def test_reference_example_programs_link_unmodified():
    """The reference's own testcases/mcmcrun{,2,3,4}.F90 (use mcmcprec / matutils / mcmcmod, external ssfunction,
    checkbounds, a user `initialize`) compile and link without a change against libmcmcxf.a + libmcmcx.so
    (oracle/Makefile, target testcases); without a GPU the program must stop loudly, never sample on the CPU."""
    import shutil, subprocess, tempfile
    if not os.path.isdir("/root/reference/testcases"):
        pytest.skip("/root/reference is only in the dev container")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "mcmcf90_amd", "fortran")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "testcases"])
    for t in ("tc_mcmcrun", "tc_mcmcrun2", "tc_mcmcrun3", "tc_mcmcrun4"):
        assert os.path.exists(os.path.join(root, "oracle", "_ref", t))
    import torch
    if torch.cuda.is_available():
        return
    with tempfile.TemporaryDirectory() as d:
        for f in ("mcmcinit.nml", "data.dat", "mcmcpar.dat", "mcmccov.dat", "mcmcsigma2.dat"):
            shutil.copy(os.path.join("/root/reference/testcases", f), d)
        p = subprocess.run([os.path.join(root, "oracle", "_ref", "tc_mcmcrun")], cwd=d, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, timeout=120)
        out = p.stdout.decode(errors="replace")
        assert p.returncode != 0 and "no HIP device" in out, out
        assert not os.path.exists(os.path.join(d, "chain.dat"))


@pytest.mark.skipif(not os.path.exists(FLANG), reason="flang not available")
def test_mcmc_main_one_without_a_gpu_stops_loudly_and_leaves_the_protocol_files_alone(oracle):
    """mcmc_main_one (MCMC_run1's one-evaluation-per-invocation protocol) goes through the engine like mcmc_main: without a
    GPU it stops with the engine's message -- no host-side fallback decides or proposes anything, mcmcrun.nml stays as the
    driver script left it and no mcmcparnew.dat appears.  (With a GPU: tests/test_gpu_run1.py.)"""
    import numpy as np
    import torch
    from oracle import refrun
    exe = os.path.join(ROOT, "oracle", "_ref", "one_shim")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/one_shim not built (make -C oracle testcases)")
    syms = subprocess.check_output(["nm", os.path.join(FDIR, "libmcmcxf.a")]).decode()
    for sym in ("mcmc_main_one_", "_QMmcmcrun1Pinit_mcmcrun_namelist", "_QMmcmcrun1Pwrite_mcmcrun_namelist"):      # mcmc_main.F90:49, mcmcrun1.F90:27,64
        assert sym in syms, sym
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by tests/test_gpu_run1.py")
    cfg = oracle.make_cfg(nsimu=5, drscale=2.0, updatesigma=0)
    prob = oracle.Problem(kind="gauss", npar=2, par0=[0.1, 0.2], cmat0=0.1 * np.eye(2), mu=np.zeros(2), lam=np.eye(2))
    with tempfile.TemporaryDirectory() as d:
        refrun.write_inputs(d, cfg, prob)
        nml = "&mcmcrun\n drstage = 1, isimu = 1, ieval = 0, nrej = 0, alpha12 = 0.0, sscrit = -1.0\n/\n"
        open(os.path.join(d, "mcmcrun.nml"), "w").write(nml)
        p = subprocess.run([exe], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
        out = p.stdout.decode(errors="replace")
        assert p.returncode != 0 and "no HIP device" in out, out
        assert open(os.path.join(d, "mcmcrun.nml")).read() == nml
        assert not os.path.exists(os.path.join(d, "mcmcparnew.dat"))


@pytest.mark.skipif(not os.path.exists(FLANG), reason="flang not available")
def test_ascii_writers_leave_the_reference_bytes():
    """the shim's writedata (matrix, vector, scalar; module matutils) against the bytes the REAL
    reference's writedata wrote for the same doubles (tests/golden/io/writedata.npz, made by oracle/gen_golden_io.py from
    oracle/_ref/wd_ref): byte for byte -- '(G0,x)' per element and a newline per row (matutils.F90:57, 889-899), '(G0)' per
    line for vectors (:950) -- and what loaddata reads back under the lock protocol is the doubles, bit for bit."""
    import sys
    import numpy as np
    sys.path.insert(0, ROOT)
    from oracle.gen_golden_io import run_probe
    subprocess.check_call(["make", "-s", "-C", FDIR])
    z = np.load(os.path.join(ROOT, "tests", "golden", "io", "writedata.npz"))
    r = run_probe(os.path.join(FDIR, "demo_writedata"), z["values"])
    assert r["mat"] == z["mat"].tobytes()
    assert r["lock"] == z["mat"].tobytes()
    assert r["vec"] == z["vec"].tobytes()
    assert r["scal"] == z["scal"].tobytes()
    assert not r["leftover"]                                             # every .lock file removed again
    assert np.array_equal(r["back"].view(np.uint64), z["values"].view(np.uint64))
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "wd_ref")):  # dev container: the reference program itself, now
        q = run_probe(os.path.join(ROOT, "oracle", "_ref", "wd_ref"), z["values"])
        assert q["mat"] == r["mat"] and q["vec"] == r["vec"] and q["scal"] == r["scal"]


@pytest.mark.skipif(not os.path.exists(FLANG), reason="flang not available")
def test_lock_file_protocol():
    """matutils.F90:1544-1680: a writer / reader under `uselock` waits while FILE.lock exists (one message a second), goes
    on when it disappears, and gives up after 10 s (status -1 to a caller that asked for it); its own lock is gone afterwards."""
    import struct
    import threading
    import time
    import numpy as np
    subprocess.check_call(["make", "-s", "-C", FDIR])
    exe = os.path.join(FDIR, "demo_writedata")
    a = np.arange(6.0).reshape(2, 3)

    def run(d):
        with open(os.path.join(d, "wd_in.bin"), "wb") as f:
            f.write(struct.pack("<2i", *a.shape)); f.write(a.astype("<f8").tobytes())
        t0 = time.time()
        p = subprocess.run([exe], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60)
        return p, time.time() - t0

    with tempfile.TemporaryDirectory() as d:                            # a lock that goes away after ~2.5 s
        lock = os.path.join(d, "wd_lock.dat.lock")
        open(lock, "w").close()
        t = threading.Timer(2.5, os.remove, args=(lock,)); t.start()
        p, dt = run(d)
        t.join()
        out = p.stdout.decode(errors="replace")
        assert p.returncode == 0, out
        assert "waiting for lock file" in out and 2.0 < dt < 9.0
        assert open(os.path.join(d, "wd_lock.dat")).read() == open(os.path.join(d, "wd_mat.dat")).read()
        assert not os.path.exists(lock)
    with tempfile.TemporaryDirectory() as d:                            # a lock that never goes away: timeout, nothing written
        lock = os.path.join(d, "wd_lock.dat.lock")
        open(lock, "w").close()
        p, dt = run(d)
        out = p.stdout.decode(errors="replace")
        assert dt > 9.5 and out.count("waiting for lock file") >= 9
        assert not os.path.exists(os.path.join(d, "wd_lock.dat"))       # writedata(..., stat, uselock) returned stat = -1
        assert os.path.exists(lock)                                     # someone else's lock is left alone
