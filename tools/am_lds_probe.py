import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmcf90_amd import engine_from_problem
from mcmcf90_amd.workloads import corr_gauss_precision
for d in (12, 16, 20):
  for n in (65536, 131072):
    for env in ("", "0"):
        if env: os.environ["MCMCX_LDS_SCRATCH"] = env
        else: os.environ.pop("MCMCX_LDS_SCRATCH", None)
        ckw = dict(nsimu=401, adaptint=100, updatesigma=0, drscale=0.0)
        pkw = dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=(0.5 / d) * np.eye(d), mu=np.zeros(d), lam=corr_gauss_precision(d))
        e = engine_from_problem(ckw, pkw, nchains=n, chain_id0=0)
        e.init(); e.run(201); e.sync()
        t0 = time.perf_counter(); e.run(401); e.sync(); dt = time.perf_counter() - t0
        print("d %d n %d LDS_SCRATCH=%s %-34s %.3g proposals/s" % (d, n, env or "default", e.last_kernel(), n * 200 / dt), flush=True)
        e.close()
