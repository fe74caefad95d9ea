"""tools/group_sweep.py -- the lane-group step kernel (mcx_group.hpp) against the lane-per-chain kernels over npar, target,
delayed rejection and chain count: proposals/s of 200 iterations between two adaptations (step kernel time only, HIP events).
GPU box.   python tools/group_sweep.py [quick]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmcf90_amd import engine_from_problem
from mcmcf90_amd.workloads import corr_gauss_precision

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
gauss_only = len(sys.argv) > 1 and sys.argv[1] == "gauss"
CASES = []
for d in ((12, 20, 32) if quick else (18, 20, 24, 28, 32) if gauss_only else (4, 8, 10, 12, 16, 20, 24, 28, 32)):
    for kind in (("gauss",) if gauss_only else ("gauss", "banana")):
        for drs in (0.0, 2.0, 3.0):
            if quick and drs == 3.0:
                continue
            CASES.append((kind, d, drs))
COUNTS = (64, 4096, 262144) if quick else (65536, 262144) if gauss_only else (64, 1024, 16384, 65536, 262144)

def rate(kind, d, drs, n, group):
    os.environ["MCMCX_GROUP"] = "1" if group else "0"
    its = 200
    ckw = dict(nsimu=100 + its + 1, adaptint=1000, updatesigma=0, drscale=drs)
    pkw = dict(kind=kind, npar=d, par0=np.zeros(d), cmat0=(0.5 / d) * np.eye(d))
    if kind == "gauss":
        pkw.update(mu=np.zeros(d), lam=corr_gauss_precision(d))
    else:
        pkw.update(b=0.1)
    e = engine_from_problem(ckw, pkw, nchains=n)
    e.init(); e.run(100); e.sync()
    e.kernel_time(reset=True)
    t0 = e.totals()
    e.run(100 + its); e.sync()
    ms, nl, ns = e.kernel_time()
    t1 = e.totals()
    k = e.last_kernel()
    e.close()
    props = n * its + (t1["drtries"] - t0["drtries"])
    return props / (ms * 1e-3), ms / its * 1e3, k

print("%-7s %4s %4s %8s | %10s %9s | %10s %9s | %5s  %s" % ("target", "npar", "drs", "chains", "lane p/s", "us/it", "group p/s", "us/it", "x", "kernels"))
for kind, d, drs in CASES:
    for n in COUNTS:
        if d > 20 and n > 65536:
            continue
        rl, ul, kl = rate(kind, d, drs, n, False)
        rg, ug, kg = rate(kind, d, drs, n, True)
        print("%-7s %4d %4.1f %8d | %10.3e %9.2f | %10.3e %9.2f | %5.2f  %s / %s" % (kind, d, drs, n, rl, ul, rg, ug, rg / rl, kl, kg), flush=True)
