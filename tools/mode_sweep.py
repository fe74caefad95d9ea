"""tools/mode_sweep.py -- one timing per mode of the engine at a common size (65536 chains; Gaussian target d = 20 unless noted):
proposals/s of iterations 201..600 (adaptint = 100: four adaptations inside).  A mode that falls off its kernels shows up
as an outlier.  Run on the GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmcf90_amd import engine_from_problem
from mcmcf90_amd.workloads import corr_gauss_precision

N = int(os.environ.get("SWEEP_CHAINS", "65536"))


def gauss(d, **kw):
    return dict(dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=0.01 * np.eye(d), mu=np.zeros(d), lam=corr_gauss_precision(d)), **kw)


d = 20
base = dict(nsimu=601, adaptint=100, updatesigma=0)
MODES = [
    ("am", dict(base, drscale=0.0), gauss(d)),
    ("am initcmatn=50", dict(base, drscale=0.0, initcmatn=50), gauss(d)),
    ("am adapthist=150 (AP)", dict(base, drscale=0.0, adapthist=150), gauss(d)),
    ("am updatesigma", dict(base, drscale=0.0, updatesigma=1, N0=1.0, S02=1.0), gauss(d, sigma2=1.0, nobs=30)),
    ("am priors+bounds", dict(base, drscale=0.0), gauss(d, pmu=np.zeros(d), psig=np.full(d, 5.0), lo=np.full(d, -8.0), hi=np.full(d, 8.0))),
    ("am condmax=1e6 (SVD factor)", dict(base, drscale=0.0, condmax=1e6), gauss(d)),
    ("dram drscale=2", dict(base, drscale=2.0), gauss(d)),
    ("dram condmax=1e6", dict(base, drscale=2.0, condmax=1e6), gauss(d)),
    ("er", dict(base, method="er"), gauss(d)),
    ("ram", dict(base, method="ram"), gauss(d)),
    ("ram condmax=1e6", dict(base, method="ram", condmax=1e6), gauss(d)),
    ("scam (d proposals per iteration)", dict(base, method="scam", nsimu=221), gauss(d)),
    ("burn-in scaling", dict(base, drscale=0.0, doburnin=1, burnintime=1000, badaptint=50, scalelimit=0.05, scalefactor=2.5), gauss(d)),
    ("burn-in greedy", dict(base, drscale=0.0, doburnin=1, burnintime=1000, badaptint=50, greedy=1, scalelimit=0.05, scalefactor=2.5), gauss(d)),
    ("banana dram", dict(base, drscale=2.0), dict(kind="banana", npar=d, par0=np.zeros(d), cmat0=0.01 * np.eye(d), b=0.1)),
    ("pooled am", dict(base, drscale=0.0), gauss(d)),
    ("pooled dram", dict(base, drscale=2.0), gauss(d)),
]
only = sys.argv[1:] 
for name, ckw, pkw in MODES:
    if only and not any(o in name for o in only):
        continue
    nsimu = ckw["nsimu"]
    n0 = 201 if nsimu > 400 else 21
    per_it = pkw["npar"] if ckw.get("method") == "scam" else 1
    try:
        e = engine_from_problem(ckw, pkw, nchains=N, chain_id0=0, pooled=1 if name.startswith("pooled") else 0)
        e.init(); e.run(n0); e.sync()
        t0 = time.perf_counter(); e.run(nsimu); e.sync(); dt = time.perf_counter() - t0
        tot = e.totals()
        print("%-36s %9.3g proposals/s  %7.1f ms  kernel %-40s accepted %.2f" % (name, N * (nsimu - n0) * per_it / dt, dt * 1e3, e.last_kernel()[:40],
                                                                              1.0 - tot["stayed"] / (N * float(nsimu - 1))), flush=True)
        e.close()
    except Exception as ex:                                    # a refused combination is a finding too
        print("%-36s FAILED: %s" % (name, str(ex)[:200]), flush=True)
