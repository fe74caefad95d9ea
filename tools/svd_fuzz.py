"""tools/svd_fuzz.py [N=24] [seed=1] -- random npar in 48..256 (odd and even), method scam / dram + condmax, adaptation intervals below
and above npar (rank-deficient covariances: the pinned Jacobi runs to its sweep cap) and random precision matrices: the adaptation's SVD
in its streamed forms (the default: 32 pair-lanes up to npar 200; MCMCX_SVD_STREAM32=0: 24; MCMCX_SVD_STREAM_B=b: b) against the
register-block form (MCMCX_SVD_STREAM=0), bit for bit on states, factors and stream positions of 70 chains.  GPU box."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmcf90_amd import engine_from_problem

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(N):
    d = int(rng.integers(48, 257))
    method = "scam" if rng.random() < 0.6 else "dram"
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    lam = A @ A.T + np.diag(10.0 ** rng.uniform(-2, 2, d))
    adaptint = int(rng.integers(4, 12)) if method == "scam" else int(rng.integers(20, 60))
    nsimu = 2 * adaptint + 2
    extra = {} if method == "scam" else dict(condmax=float(10.0 ** rng.uniform(1, 8)))
    if method == "dram" and rng.random() < 0.3:
        extra["drscale"] = 2.0
    ckw = dict(nsimu=nsimu, method=method, adaptint=adaptint, updatesigma=0, **extra)
    pkw = dict(kind="gauss", npar=d, par0=np.full(d, 0.05), cmat0=(0.3 / d) * np.eye(d), mu=np.zeros(d), lam=lam)
    bsel = int(rng.integers(2, 25))
    res = {}
    for path, env in (("stream", {}), ("stream24", {"MCMCX_SVD_STREAM32": "0"}), ("streamb", {"MCMCX_SVD_STREAM_B": str(bsel)}), ("reg", {"MCMCX_SVD_STREAM": "0"})):
        for k in ("MCMCX_SVD_STREAM32", "MCMCX_SVD_STREAM_B", "MCMCX_SVD_STREAM"):
            os.environ.pop(k, None)
        os.environ.update(env)
        e = engine_from_problem(ckw, pkw, nchains=70, chain_id0=3, record_accept=1)
        e.init(); e.run()
        res[path] = (e.theta().view(np.uint64), e.accept_masks(), [np.asarray(e.R(c)).view(np.uint64) for c in (0, 63, 69)], [e.rng(c)[0] for c in (0, 69)])
        e.close()
    a = res["reg"]
    ok = True
    for path in ("stream", "stream24", "streamb"):
        b = res[path]
        ok = ok and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and all(np.array_equal(x, y) for x, y in zip(a[2], b[2])) and a[3] == b[3]
    bad += 0 if ok else 1
    print("case %2d npar %3d %-4s adaptint %2d b %2d %s: %s" % (case, d, method, adaptint, bsel, extra, "equal" if ok else "DIFFERENT"), flush=True)
print("%d configurations, %d differences" % (N, bad))
sys.exit(1 if bad else 0)
