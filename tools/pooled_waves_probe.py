"""tools/pooled_waves_probe.py -- pooled AM on the matrix cores (pooled_mfma_kernel<false>) with one and with two waves per SIMD (MCMCX_POOLED_WAVES)
over npar x chain count: proposals/s of 200 iterations.  GPU box."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from mcmcf90_amd import engine_from_problem
from mcmcf90_amd.workloads import corr_gauss_precision
for d in (10, 20, 32, 50):
    for n in (65536, 131072, 262144, 1048576):
        lam = corr_gauss_precision(d)
        ckw = dict(nsimu=302, method="dram", adaptint=100, updatesigma=0, drscale=0.0)
        pkw = dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=0.01 * np.eye(d), mu=np.zeros(d), lam=lam)
        r = []
        for w in ("1", "2"):
            os.environ["MCMCX_POOLED_WAVES"] = w
            e = engine_from_problem(ckw, pkw, nchains=n, pooled=1)
            e.init(); e.run(101); e.sync()
            t0 = time.perf_counter(); e.run(301); e.sync(); dt = time.perf_counter() - t0
            r.append(n * 200 / dt); k = e.last_kernel(); e.close()
        print("npar %3d %8d chains (%5d tiles): one wave %.3e  two waves %.3e  x%.2f  %s" % (d, n, n // 64, r[0], r[1], r[1] / r[0], k), flush=True)
