"""tools/pooled_ks_tiles.py -- from how many tiles on the forty-row form (pooled_mfma_ks_kernel, MCMCX_POOLED_KS=1) beats what the engine takes without it
(MCMCX_POOLED_KS=0: pooled_mfma_kernel<false> below 8192 tiles at npar > 40, <false, true> from there on): the threshold in pooled_forty_rows
(mcx_host_launch.hpp).  GPU box."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmcf90_amd import engine_from_problem
from mcmcf90_amd.workloads import corr_gauss_precision
os.environ.pop("MCMCX_POOLED_WAVES", None)
print("%-5s %-7s %12s %12s %8s  %s" % ("npar", "tiles", "forty rows", "without", "ratio", "(kernel ms per 100 iterations; the kernel taken without)"), flush=True)
for d in (45, 50, 64):
    for tiles in (1024, 2048, 4096, 8192, 16384):
        ms = {}
        for ks in ("1", "0"):
            os.environ["MCMCX_POOLED_KS"] = ks
            c = dict(nsimu=301, method="ram", adaptint=100, updatesigma=0)
            p = dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=0.01 * np.eye(d), mu=np.zeros(d), lam=corr_gauss_precision(d))
            e = engine_from_problem(c, p, nchains=64 * tiles, pooled=1)
            e.init(); e.run(100); e.sync(); e.kernel_time(reset=True)
            e.run(200); e.sync()
            t, nl, _ = e.kernel_time()
            ms[ks] = (t / max(nl, 1), e.last_kernel())
            e.close()
        assert ms["1"][1] == "pooled_mfma_ks_kernel", ms
        print("%-5d %-7d %12.3f %12.3f %8.3f  %s" % (d, tiles, ms["1"][0], ms["0"][0], ms["0"][0] / ms["1"][0], ms["0"][1]), flush=True)
