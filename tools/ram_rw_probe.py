"""tools/ram_rw_probe.py -- method=ram over npar x start regime, proposals/s of 200 iterations (MCMCX_LIBRARY=<variant> for an A/B build, e.g. -DMCX_RW=10).  GPU box."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from mcmcf90_amd import engine_from_problem
from mcmcf90_amd.workloads import corr_gauss_precision
for d in (12, 20, 32, 50, 64, 100):
    for start in ("default", "target"):
        lam = corr_gauss_precision(d)
        cm = 0.01 * np.eye(d) if start == "default" else np.linalg.inv(lam)
        ckw = dict(nsimu=302, method="ram", adaptint=100, updatesigma=0)
        pkw = dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=cm, mu=np.zeros(d), lam=lam)
        n = 131072 if d <= 64 else 32768
        e = engine_from_problem(ckw, pkw, nchains=n)
        e.init(); e.run(101); e.sync()
        t0 = time.perf_counter(); e.run(301); e.sync(); dt = time.perf_counter() - t0
        print("npar %3d %-7s %6d chains: %.3e proposals/s  %s" % (d, start, n, n * 200 / dt, e.last_kernel()), flush=True)
        e.close()
