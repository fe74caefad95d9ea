"""tools/dr_sweep.py -- delayed rejection with per-chain factors over npar: step_kernel_dr (two LDS vectors per wave: 160 KiB / (npar KiB)
waves per CU) against step_kernel_dr_big (vectors in global scratch, MCMCX_DR_BIG=1).  Run on the GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmcf90_amd import engine_from_problem
from mcmcf90_amd.workloads import corr_gauss_precision

for d, n in [(int(a), int(b)) for a, b in (x.split(":") for x in os.environ.get("DR_SWEEP", "20:131072,40:131072,64:65536,80:65536,100:65536,130:32768,160:32768").split(","))]:
    for big in (0, 1):
        os.environ["MCMCX_DR_BIG"] = str(big)
        ckw = dict(nsimu=401, adaptint=100, updatesigma=0, drscale=2.0)
        pkw = dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=(0.5 / d) * np.eye(d), mu=np.zeros(d), lam=corr_gauss_precision(d))
        e = engine_from_problem(ckw, pkw, nchains=n, chain_id0=0)
        e.init(); e.run(201); e.sync()
        t0 = time.perf_counter(); e.run(401); e.sync(); dt = time.perf_counter() - t0
        print("npar %3d  %6d chains  %-22s %9.3g iterations/s  %7.1f ms" % (d, n, e.last_kernel(), n * 200 / dt, dt * 1e3), flush=True)
        e.close()
