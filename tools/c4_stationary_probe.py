"""Throughput of the RAM step kernel when the chains are AT the target acceptance rate (tools, not the bench):
config 4's target started from cmat0 = Sigma (MCMC_init scales the factor by 2.4/sqrt(d), MCMC_init.F90:109), so that
alpha scatters around alphatarget and most iterations are Cholesky downdates (the bench's own start, cmat0 = 0.01 I, accepts 86 % and adapts
almost only by updates).   python tools/c4_stationary_probe.py [nchains=131072] [its=300]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()
from mcmcf90_amd import engine_from_problem
from mcmcf90_amd.workloads import problem

nch = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
its = int(sys.argv[2]) if len(sys.argv) > 2 else 300
for label, scale in (("bench start (cmat0 = 0.01 I)", None), ("stationary start (cmat0 = Sigma)", 1.0)):
    ckw, pkw, _ = problem("c4", its + 1)
    if scale is not None:
        pkw = dict(pkw, cmat0=scale * np.linalg.inv(pkw["lam"]))
    e = engine_from_problem(ckw, pkw, nchains=nch)
    e.init(); e.run(101); e.sync(); e.kernel_time(reset=True)
    t0 = time.perf_counter(); e.run(its + 1); e.sync(); dt = time.perf_counter() - t0
    tot = e.totals()
    print("%s: %.3g proposals/s, stayed %.3f" % (label, nch * (its - 100) / dt, tot["stayed"] / (nch * its)))
    e.close()
