"""tools/pooled_ks_sweep.py -- pooled_mfma_ks_kernel against pooled_mfma_kernel<false, true> over npar 41..64 at 1 048 576 chains: where the forty-row form pays
(profiles/r06_j/pooled_ks_sweep.txt; the threshold over tiles: tools/pooled_ks_tiles.py).  GPU box."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmcf90_amd import engine_from_problem
from mcmcf90_amd.workloads import corr_gauss_precision
n = 1048576
print("%-5s %-22s %12s %12s %8s   (kernel ms per 100 iterations: forty rows, library, ratio)" % ("npar", "case", "ks", "library", "lib/ks"), flush=True)
for d in (41, 44, 45, 48, 50, 52, 53, 56, 60, 64):
    for name, ckw, kind in (("gauss ram", dict(method="ram", adaptint=100), "gauss"), ("gauss am sigma2", dict(adaptint=100, updatesigma=1), "gauss"),
                            ("banana am", dict(adaptint=100), "banana")):
        if kind == "banana" and d not in (41, 50, 64):
            continue
        ms = {}
        for ks in ("1", "0"):
            os.environ["MCMCX_POOLED_WAVES"] = "2"; os.environ["MCMCX_POOLED_KS"] = ks
            c = dict(dict(nsimu=301, updatesigma=0, N0=1.0, S02=0.5), **ckw)
            if kind == "gauss":
                p = dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=0.01 * np.eye(d), mu=np.zeros(d), lam=corr_gauss_precision(d))
                if c.get("updatesigma"):
                    p.update(sigma2=0.8, nobs=20)
            else:
                p = dict(kind="banana", npar=d, par0=np.zeros(d), cmat0=(1.0 / d) * np.eye(d), b=0.1)
            e = engine_from_problem(c, p, nchains=n, pooled=1)
            e.init(); e.run(100); e.sync(); e.kernel_time(reset=True)
            e.run(200); e.sync()
            t, nl, _ = e.kernel_time()
            ms[ks] = (t / max(nl, 1), e.last_kernel())
            e.close()
        assert ms["1"][1] == "pooled_mfma_ks_kernel" and ms["0"][1] == "pooled_mfma_kernel<false, true>", ms
        print("%-5d %-22s %12.3f %12.3f %8.3f" % (d, name, ms["1"][0], ms["0"][0], ms["0"][0] / ms["1"][0]), flush=True)
