"""oracle/refrun.py -- TEST INFRASTRUCTURE. Not part of the product path.

Runs the REAL mcmcf90 reference (oracle/_ref/mcxref, built by oracle/Makefile from
/root/reference) in a scratch directory on a Problem/config pair and parses what
MCMC_writechains (MCMC_aux.F90:17-85) leaves behind.  The uniform stream is the
pinned Philox stream (oracle/ref/rng_interpose.c), keyed by (seed, chain_id).
"""
import os
import shutil
import struct
import subprocess
import tempfile
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(_HERE, "_ref", "mcxref")
EXE_SVD = os.path.join(_HERE, "_ref", "mcxref_svd")      # same program, dgesvd = the pinned Jacobi routine
EXE_MKLLOG = os.path.join(_HERE, "_ref", "mcxref_mkllog")  # same program, MKL's dgesvd, every call logged (ref/dgesvd_logger.c)
METHOD_NAMES = {0: "dram", 1: "ram", 2: "scam", 3: "er"}
TARGET_IDS = {"gauss": 0, "banana": 1, "expdata": 2}


def available():
    return os.path.exists(EXE)


def read_mat4(path):
    """MAT-v4 file written by matfiles.F90:66-126: 5 x int32 header, name, column-major doubles."""
    with open(path, "rb") as f:
        raw = f.read()
    out, off = {}, 0
    while off < len(raw):
        mopt, mrows, ncols, imagf, namlen = struct.unpack_from("<5i", raw, off)
        off += 20
        name = raw[off:off + namlen].split(b"\0")[0].decode()
        off += namlen
        assert mopt == 0 and imagf == 0, (mopt, imagf)
        a = np.frombuffer(raw, dtype="<f8", count=mrows * ncols, offset=off).reshape(ncols, mrows).T.copy()
        off += 8 * mrows * ncols
        out[name] = a
    return out


def _fdbl(v):
    """Fortran double-precision literal with 17 significant digits."""
    t = "%.17g" % float(v)
    return t.replace("e", "d") if "e" in t else t + "d0"


def _hex(v):
    return " ".join(float(x).hex() for x in np.asarray(v, dtype=np.float64).ravel())


def _dat(path, a):
    a = np.atleast_2d(np.asarray(a, dtype=np.float64))
    with open(path, "w") as f:
        for row in a:
            f.write(" ".join(repr(float(x)) for x in row) + "\n")


def write_inputs(d, cfg, prob, extra_nml=""):
    fields = ["nsimu", "doadapt", "doburnin", "adaptint", "adapthist", "badaptint", "adaptend", "initcmatn",
              "burnintime", "greedy", "updatesigma"]
    dfields = ["scalelimit", "scalefactor", "drscale", "N0", "S02", "condmax", "alphatarget", "nuparam"]
    with open(os.path.join(d, "mcmcinit.nml"), "w") as f:
        f.write("&mcmc\n")
        f.write(" method = '%s'\n" % METHOD_NAMES[cfg.method])
        for k in fields:
            f.write(" %s = %d\n" % (k, getattr(cfg, k)))
        for k in dfields:
            f.write(" %s = %s\n" % (k, _fdbl(getattr(cfg, k))))
        f.write(" verbosity = 0\n printint = 100000000\n")
        f.write(" chainfile = 'chain.mat'\n ssfile = 'sschain.mat'\n s2file = 's2chain.mat'\n")
        if prob.pri_mu is not None:
            f.write(" priorsfile = 'priors.dat'\n")
        f.write(extra_nml)
        f.write("/\n")
    _dat(os.path.join(d, "mcmcpar.dat"), prob.par0.reshape(1, -1))
    _dat(os.path.join(d, "mcmccov.dat"), prob.cmat0)
    ny = getattr(prob, "ny", 1)
    _dat(os.path.join(d, "mcmcsigma2.dat"), np.vstack([prob.sigma2v, prob.nobsv.astype(np.float64)]))     # 2 x nycol
    if ny > 1:
        _dat(os.path.join(d, "mcmcnycol.dat"), np.array([[float(ny)]]))                               # initialize.F90:52-58
    if prob.pri_mu is not None:
        _dat(os.path.join(d, "priors.dat"), np.vstack([prob.pri_mu, prob.pri_sig]))
    with open(os.path.join(d, "mcxtarget.txt"), "w") as f:
        f.write("%d %d\n" % (3 if ny > 1 else TARGET_IDS[prob.kind], prob.npar))
        if prob.kind == "gauss":
            f.write(_hex(prob.mu) + "\n" + _hex(prob.lam) + "\n")
        elif prob.kind == "banana":
            f.write(_hex([prob.b]) + "\n")
        elif ny > 1:
            f.write("%d %d\n" % (ny, len(prob.xdata)) + _hex(prob.xdata) + "\n" + _hex(prob.ydata) + "\n")
        else:
            f.write("%d\n" % len(prob.xdata) + _hex(prob.xdata) + "\n" + _hex(prob.ydata) + "\n")
        if prob.lo is not None or prob.hi is not None:
            lo = prob.lo if prob.lo is not None else np.full(prob.npar, -np.inf)
            hi = prob.hi if prob.hi is not None else np.full(prob.npar, np.inf)
            f.write("%d\n" % prob.npar + _hex(lo) + "\n" + _hex(hi) + "\n")
        else:
            f.write("0\n")


class RefResult:
    pass


def read_svd_log(path):
    """[(info, s[n], U[n, n])] of every dgesvd call logged by ref/dgesvd_logger.c"""
    calls = []
    if not os.path.exists(path):
        return calls
    raw = open(path, "rb").read()
    off = 0
    while off < len(raw):
        n, info = struct.unpack_from("<2i", raw, off); off += 8
        sv = np.frombuffer(raw, dtype="<f8", count=n, offset=off).copy(); off += 8 * n
        U = np.frombuffer(raw, dtype="<f8", count=n * n, offset=off).reshape(n, n).T.copy(); off += 8 * n * n
        calls.append((info, sv, U))
    return calls


def run_reference(cfg, prob, seed=0x6D636D63, chain_id=0, keep=False, timeout=600, pinned_svd=False, timing_only=False, svd_log=False):
    """timing_only: run in a tmpfs scratch directory when there is one, time the reference process alone (r.seconds)
    and do not parse its output files (bench.py's cpu_baseline leg)."""
    if not available():
        raise RuntimeError("oracle/_ref/mcxref not built (make -C oracle ref)")
    import time
    shm = "/dev/shm" if (timing_only and os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK)) else None
    d = tempfile.mkdtemp(prefix="mcxref_", dir=shm)
    try:
        write_inputs(d, cfg, prob)
        env = dict(os.environ, MCX_SEED=str(seed), MCX_CHAIN=str(chain_id), MKL_NUM_THREADS="1",
                   MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="1", MCX_RNG_LOG=os.path.join(d, "rng.log"))
        t0 = time.perf_counter()
        if svd_log:
            env["MCX_SVD_LOG"] = os.path.join(d, "svd.log")
        exe = EXE_MKLLOG if svd_log else (EXE_SVD if pinned_svd else EXE)
        p = subprocess.run([exe], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
        r = RefResult()
        if svd_log:
            r.svd_calls = read_svd_log(env["MCX_SVD_LOG"])
        r.seconds = time.perf_counter() - t0
        r.scratch = "tmpfs" if shm else "disk"
        r.stdout = p.stdout.decode(errors="replace")
        r.returncode = p.returncode
        if not os.path.exists(os.path.join(d, "chain.mat")):
            raise RuntimeError("reference run produced no chain:\n" + r.stdout[-2000:])
        if timing_only:
            return r
        r.chain = read_mat4(os.path.join(d, "chain.mat"))["chain"]
        r.sschain = read_mat4(os.path.join(d, "sschain.mat"))["sschain"]
        r.s2chain = read_mat4(os.path.join(d, "s2chain.mat"))["s2chain"] if cfg.updatesigma else None
        if r.s2chain is not None and r.s2chain.shape[1] == 1:
            r.s2chain = r.s2chain[:, 0]
        r.chaincmat = np.loadtxt(os.path.join(d, "mcmccovf.dat"), ndmin=2)
        r.chainmean = np.loadtxt(os.path.join(d, "mcmcmean.dat"), ndmin=1)
        r.rng_n = int(open(os.path.join(d, "rng.log")).read().split()[-1])
        r.chainind = r.chain.shape[0]
        return r
    finally:
        if keep:
            print("kept", d)
        else:
            shutil.rmtree(d, ignore_errors=True)


def run_reference_many(cfg, prob, nprocs, seed=0x6D636D63, chain_id0=0, timeout=300, pinned_svd=False):
    """nprocs processes of the reference at once (bench.py's all-cores leg): process c runs the chain keyed
    (seed, chain_id0 + c) in its own scratch directory (tmpfs when there is one).  Timing only: returns
    {"wall": seconds from the first start to the last exit, "cpu_seconds": user + system time of all of them, "scratch"}."""
    if not available():
        raise RuntimeError("oracle/_ref/mcxref not built (make -C oracle ref)")
    import time
    shm = "/dev/shm" if (os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK)) else None
    top = tempfile.mkdtemp(prefix="mcxrefN_", dir=shm)
    exe = EXE_SVD if pinned_svd else EXE
    procs = []
    try:
        dirs = []
        for c in range(nprocs):
            d = os.path.join(top, "p%d" % c)
            os.mkdir(d)
            if c == 0:
                write_inputs(d, cfg, prob)
            else:
                for f in os.listdir(dirs[0]):
                    os.link(os.path.join(dirs[0], f), os.path.join(d, f))       # inputs are read-only: hard links, no copies
            dirs.append(d)
        base = dict(os.environ, MCX_SEED=str(seed), MKL_NUM_THREADS="1", MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="1")
        t0 = time.perf_counter()
        for c, d in enumerate(dirs):
            procs.append(subprocess.Popen([exe], cwd=d, env=dict(base, MCX_CHAIN=str(chain_id0 + c)),
                                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
        cpu = 0.0
        left = {p.pid: p for p in procs}
        while left:
            if time.perf_counter() - t0 > timeout:
                raise RuntimeError("%d of %d reference processes still running after %d s" % (len(left), nprocs, timeout))
            pid, status, ru = os.wait4(-1, os.WNOHANG)
            if pid == 0:
                time.sleep(0.002)
                continue
            p = left.pop(pid, None)
            if p is None:
                continue                                   # some other child of the caller
            p.returncode = os.waitstatus_to_exitcode(status) if hasattr(os, "waitstatus_to_exitcode") else (status >> 8)
            cpu += ru.ru_utime + ru.ru_stime
            if p.returncode != 0:
                raise RuntimeError("a reference process exited with code %d" % p.returncode)
        wall = time.perf_counter() - t0
        for d in dirs:
            if not os.path.exists(os.path.join(d, "chain.mat")):
                raise RuntimeError("a reference process produced no chain")
        return {"wall": wall, "cpu_seconds": cpu, "scratch": "tmpfs" if shm else "disk"}
    finally:
        for p in procs:
            if p.returncode is None:
                try:
                    p.kill()
                except OSError:
                    pass
        shutil.rmtree(top, ignore_errors=True)


def accepted_from_chain(chain, nsimu):
    """Expand the run-length column (MCMC_aux.F90:167-175) to the per-iteration accept flags."""
    cnt = chain[:, -1].astype(np.int64)
    acc = np.zeros(int(cnt.sum()), dtype=np.uint8)
    acc[np.concatenate([[0], np.cumsum(cnt)[:-1]])] = 1
    assert len(acc) == nsimu, (len(acc), nsimu)
    return acc


# ---------------------------------------------------------------- mcmc_main_one: one invocation at a time
EXE_ONE = os.path.join(_HERE, "_ref", "mcxref_one")     # same callbacks, main program calls mcmc_main_one (MCMC_run1[_er])
EXE_ONE_SVD = os.path.join(_HERE, "_ref", "mcxref_one_svd")   # ... with dgesvd = the pinned Jacobi routine (condmax > 0)


def _nums(path):
    return np.array([float(t) for t in open(path).read().split()], dtype=np.float64)


def read_run1_files(d):
    """what one invocation of mcmc_main_one leaves in directory d, in the dict layout of oracle/run1.py"""
    import re
    txt = open(os.path.join(d, "mcmcrun.nml")).read().upper().replace("\n", " ")
    f = {}
    for k, conv in (("DRSTAGE", int), ("ISIMU", int), ("IEVAL", int), ("NREJ", int), ("ALPHA12", float), ("SSCRIT", float)):
        m = re.search(k + r"\s*=\s*([-+0-9.EDed]+)", txt)
        f[k.lower()] = conv(m.group(1).replace("D", "E").rstrip(",")) if conv is float else int(m.group(1).rstrip(",").rstrip("."))
    for key, name in (("mean", "mcmcmean.dat"), ("parf", "mcmcparf.dat"), ("oldpar1", "mcmcoldpar1.dat"), ("oldpar2", "mcmcoldpar2.dat"),
                      ("ssprev1", "mcmcssprev1.dat"), ("ssprev2", "mcmcssprev2.dat"), ("parnew", "mcmcparnew.dat"),
                      ("sscritfile", "mcmcsscrit.dat")):
        p = os.path.join(d, name)
        f[key] = _nums(p) if os.path.exists(p) else None
    if f["sscritfile"] is not None:
        f["sscritfile"] = float(f["sscritfile"][0])
    acc, rej = os.path.exists(os.path.join(d, "mcmc_accepted")), os.path.exists(os.path.join(d, "mcmc_rejected"))
    assert acc != rej, "exactly one of mcmc_accepted / mcmc_rejected must exist"
    f["accepted"] = acc
    f["done"] = os.path.exists(os.path.join(d, "mcmc_run_done"))
    return f


def run_program_one(exe, cfg, prob, seeds, extra_nml="", seedfile=False, keep=False):
    """Drive a mcmc_main_one program (the reference's mcxref_one, or the same user program linked against the engine's
    shim) through len(seeds) invocations the way the protocol's driver script would: mcmcrun.nml initialised once
    (init_mcmcrun_namelist), then run, copy mcmcparnew.dat over mcmcpar.dat, run again.  Invocation k draws from the
    stream keyed (seeds[k], 0): through MCX_SEED for the interposed reference, through the seed file for the shim.
    Returns the list of files dicts."""
    d = tempfile.mkdtemp(prefix="mcxone_")
    out = []
    try:
        write_inputs(d, cfg, prob, extra_nml=extra_nml)
        with open(os.path.join(d, "mcmcrun.nml"), "w") as f:
            f.write("&mcmcrun\n drstage = 1, isimu = 1, ieval = 0, nrej = 0, alpha12 = 0.0, sscrit = -1.0\n/\n")
        for s in seeds:
            env = dict(os.environ, MCX_SEED=str(s), MCX_CHAIN="0", MKL_NUM_THREADS="1", MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="1")
            if seedfile:
                open(os.path.join(d, "gfortran_seed.dat"), "w").write("%d\n" % s)
            p = subprocess.run([exe], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
            if p.returncode != 0 or not os.path.exists(os.path.join(d, "mcmcparnew.dat")):
                raise RuntimeError("mcmc_main_one failed:\n" + p.stdout.decode(errors="replace")[-3000:])
            f = read_run1_files(d)
            f["stdout"] = p.stdout.decode(errors="replace")
            f["chainrow"] = (read_mat4(os.path.join(d, "chain.mat"))["chain"] if os.path.exists(os.path.join(d, "chain.mat"))
                             else np.loadtxt(os.path.join(d, "chain.dat"), ndmin=2))[0]
            out.append(f)
            shutil.copy(os.path.join(d, "mcmcparnew.dat"), os.path.join(d, "mcmcpar.dat"))
        return out
    finally:
        if keep:
            print("kept", d)
        else:
            shutil.rmtree(d, ignore_errors=True)
