import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmcf90_amd import engine_from_problem
from mcmcf90_amd.workloads import corr_gauss_precision
for d, n in ((20, 131072), (50, 131072)):
    for scalar in ("0", "1"):
        os.environ["MCMCX_POOLED_SCALAR"] = scalar
        ckw = dict(nsimu=401, adaptint=100, updatesigma=0, method="er")
        pkw = dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=(0.5 / d) * np.eye(d), mu=np.zeros(d), lam=corr_gauss_precision(d))
        e = engine_from_problem(ckw, pkw, nchains=n, chain_id0=0, pooled=1)
        e.init(); e.run(201); e.sync()
        t0 = time.perf_counter(); e.run(401); e.sync(); dt = time.perf_counter() - t0
        print("pooled er npar %3d  %-34s %9.3g iterations/s" % (d, e.last_kernel()[:34], n * 200 / dt), flush=True)
        e.close()
