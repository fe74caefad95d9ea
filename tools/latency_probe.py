"""tools/latency_probe.py -- time per iteration of ONE tile (64 chains) over npar and method: the latency of a single wave's
iteration, which is what bounds runs with few chains.  GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmcf90_amd import engine_from_problem
from mcmcf90_amd.workloads import corr_gauss_precision
for method, extra in (("dram", dict(drscale=0.0)), ("ram", {}), ("dram", dict(drscale=2.0))):
    for d in (2, 5, 10, 20, 50):
        if d > 32 and extra.get("drscale", 0.0) > 0.0 and os.environ.get("MCMCX_GROUP") != "0":
            pass                                   # (delayed rejection above npar 32 stays on the lane kernels)
        ckw = dict(nsimu=2001, adaptint=100, updatesigma=0, method=method, **extra)
        pkw = dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=(0.5 / d) * np.eye(d), mu=np.zeros(d), lam=corr_gauss_precision(d))
        e = engine_from_problem(ckw, pkw, nchains=64, chain_id0=0)
        e.init(); e.run(1001); e.sync()
        t0 = time.perf_counter(); e.run(2001); e.sync(); dt = time.perf_counter() - t0
        print("%-4s drscale %.0f npar %3d  %-32s %7.2f us per iteration" % (method, extra.get("drscale", 0.0), d, e.last_kernel()[:32], dt / 1000 * 1e6), flush=True)
        e.close()
