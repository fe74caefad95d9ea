"""tools/pooled_dr_fuzz.py [first] [last] -- random pooled configurations with delayed rejection or early rejection: the matrix-core
kernel (pooled_mfma_kernel<DR>) against the lane-per-chain kernels (MCMCX_POOLED_SCALAR=1), bit for bit: states, accept masks,
stream positions, counters, the shared factor and the second-stage tables.  GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmcf90_amd import engine_from_problem

A = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 120
bad = []; t0 = time.time()
for seed in range(A, B):
    r = np.random.default_rng(7000 + seed)
    d = int(r.choice([21, 24, 31, 32, 33, 47, 48, 50, 63, 64, 65, 80, 100, 129, 140]))
    N = int(r.choice([66, 130, 200]))
    er = r.random() < 0.25
    ckw = dict(nsimu=int(r.integers(120, 260)), adaptint=int(r.choice([40, 100])), updatesigma=int(r.integers(0, 2)))
    if er: ckw["method"] = "er"
    else: ckw["drscale"] = float(r.choice([1.5, 2.0, 3.0]))
    if r.random() < 0.3: ckw.update(doburnin=1, burnintime=int(r.integers(50, 120)), badaptint=int(r.choice([25, 50])), scalelimit=float(r.choice([0.2, 0.4])), scalefactor=2.5)
    if r.random() < 0.25: ckw["condmax"] = float(r.choice([1e6, 50.0]))
    kind = "banana" if (r.random() < 0.3 and d <= 64) else "gauss"
    c0 = float(r.choice([0.02, 0.3, 2.0])) / d
    if kind == "gauss":
        Aa = r.standard_normal((d, d)) / np.sqrt(d)
        pkw = dict(kind="gauss", npar=d, par0=np.full(d, 0.1), cmat0=c0 * np.eye(d), mu=np.linspace(-1, 1, d), lam=Aa @ Aa.T + np.eye(d))
    else:
        pkw = dict(kind="banana", npar=d, par0=np.zeros(d), cmat0=c0 * np.eye(d), b=0.1)
    if r.random() < 0.3: pkw.update(lo=np.full(d, -2.5), hi=np.full(d, 2.5))
    if r.random() < 0.3: pkw.update(pri_mu=np.zeros(d), pri_sig=np.where(np.arange(d) % 3 == 0, 0.0, 1.5))
    if ckw["updatesigma"]: pkw.update(sigma2=0.8, nobs=25)
    res = []
    try:
        for scalar in ("0", "1"):
            os.environ["MCMCX_POOLED_SCALAR"] = scalar
            e = engine_from_problem(ckw, pkw, nchains=N, pooled=1, record_accept=1, chain_id0=3)
            e.init(); e.run()
            tot = e.totals()
            res.append((e.last_kernel(), e.theta().copy(), e.accept_masks().copy(), [e.rng(c)[0] for c in (0, 65, N - 1)], e.pooled()[3].copy(),
                        (tot["stayed"], tot["drtries"], tot["draccepted"], tot["bndstayed"])))
            e.close()
        a, b = res
        ok = (np.array_equal(a[1].view(np.uint64), b[1].view(np.uint64)) and np.array_equal(a[2], b[2]) and a[3] == b[3]
              and np.array_equal(a[4].view(np.uint64), b[4].view(np.uint64)) and a[5] == b[5])
        if not ok or "mfma" not in a[0]:
            bad.append((seed, d, N, ckw, kind, a[0], b[0], ok))
    except Exception as ex:
        bad.append((seed, d, N, ckw, kind, repr(ex)[:200]))
    if seed % 20 == 0:
        print("seed", seed, "failures", len(bad), "%.0f s" % (time.time() - t0), flush=True)
print("configurations", B - A, "failures", len(bad))
for x in bad[:10]: print(x)
