"""tools/pooled_ks_check.py -- pooled_mfma_ks_kernel (the tile's LDS vector in two pieces of forty rows: eight tiles per CU at npar 41..64,
MCMCX_POOLED_KS=1) against pooled_mfma_kernel<false, true> (MCMCX_POOLED_KS=0) on the same 77 configurations, bit for bit -- states, accept
ballots, stream positions, scalars, the pooled factor -- and both timed at BASELINE config 4's size.  The kernel's development harness (it was a
tools/variants form until it won: docs/history/r06.md section 8c); the suite's own cases are tests/test_gpu_pooled.py.  GPU box."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmcf90_amd import engine_from_problem
from mcmcf90_amd.workloads import corr_gauss_precision


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def run(ckw, pkw, n, ks, cut=None):
    os.environ["MCMCX_POOLED_WAVES"] = "2"
    os.environ["MCMCX_POOLED_KS"] = "1" if ks else "0"
    e = engine_from_problem(ckw, pkw, nchains=n, pooled=1, record_accept=1)
    e.init()
    if cut:
        e.run(cut)
    e.run()
    k = e.last_kernel()
    out = (e.theta().copy(), e.accept_masks().copy(), [e.rng(c) for c in (0, 63, n - 1)], e.scalars().copy(), e.pooled()[3].copy(), e.totals())
    e.close()
    return k, out


ok = True
r = np.random.default_rng(7)
cases = []
for d in (41, 44, 45, 48, 49, 50, 53, 57, 60, 63, 64):
    for kind, extra in (("gauss", {}), ("gauss", dict(updatesigma=1)), ("banana", {}), ("gauss", dict(method="er")), ("gauss", dict(method="ram", adaptint=20)),
                        ("gauss", dict(condmax=1e8)), ("gauss", dict(doburnin=1, burnintime=60, badaptint=25, scalelimit=0.3))):
        cases.append((d, kind, extra))
for d, kind, extra in cases:
    n = int(r.choice([64, 70, 130, 200]))
    A = r.standard_normal((d, d)) / np.sqrt(d)
    ckw = dict(dict(nsimu=130, adaptint=50, updatesigma=0, N0=1.0, S02=0.5), **extra)
    if kind == "gauss":
        pkw = dict(kind="gauss", npar=d, par0=np.full(d, 0.1), cmat0=(0.3 / d) * np.eye(d), mu=np.linspace(-1, 1, d), lam=A @ A.T + np.eye(d))
        if ckw.get("updatesigma"):
            pkw.update(sigma2=0.8, nobs=20)
        if r.random() < 0.4:
            pkw.update(lo=np.full(d, -1.5), hi=np.full(d, 1.5))
        if r.random() < 0.4:
            pkw.update(pri_mu=np.zeros(d), pri_sig=np.where(np.arange(d) % 3 == 0, 0.0, 2.0))
    else:
        pkw = dict(kind="banana", npar=d, par0=np.zeros(d), cmat0=(1.0 / d) * np.eye(d), b=0.1)
    cut = int(r.choice([0, 57]))
    ka, a = run(ckw, pkw, n, True, cut or None)
    kb, b = run(ckw, pkw, n, False, cut or None)
    same = (ka == "pooled_mfma_ks_kernel" and kb == "pooled_mfma_kernel<false, true>" and np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(a[1], b[1])
            and a[2] == b[2] and np.array_equal(bits(a[3]), bits(b[3])) and np.array_equal(bits(a[4]), bits(b[4])) and a[5] == b[5])
    ok &= same
    if not same:
        print("DIFFERS: npar %d %s %s n %d cut %d kernels %s / %s" % (d, kind, extra, n, cut, ka, kb), flush=True)
print("%d configurations: %s" % (len(cases), "all bit-equal" if ok else "DIFFERENCES"), flush=True)

# timing at config 4's size
for rep in (1, 2):
    for ks in (True, False):
        d, n = 50, 1048576
        os.environ["MCMCX_POOLED_WAVES"] = "2"
        os.environ["MCMCX_POOLED_KS"] = "1" if ks else "0"
        ckw = dict(nsimu=401, method="ram", updatesigma=0, adaptint=100)
        pkw = dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=0.01 * np.eye(d), mu=np.zeros(d), lam=corr_gauss_precision(d))
        e = engine_from_problem(ckw, pkw, nchains=n, pooled=1)
        e.init(); e.run(100); e.sync(); e.kernel_time(reset=True)
        t0 = time.perf_counter(); e.run(400); e.sync(); dt = time.perf_counter() - t0
        ms, nl, ns = e.kernel_time()
        print("npar 50, 1048576 chains: %-34s %.3f ms per 100 iterations (kernel %.3f)  %.4g proposals/s" % (e.last_kernel(), dt / 3 * 1e3, ms / max(nl, 1), n * 300 / dt), flush=True)
        e.close()
sys.exit(0 if ok else 1)
