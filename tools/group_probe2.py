"""tools/group_probe2.py -- lane-group against lane-per-chain step kernels where tools/group_sweep.py does not look: npar 40..64 without
delayed rejection (BASELINE config 4's target with method = 'dram'), and the sigma2 update.  GPU box."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmcf90_amd import engine_from_problem
from mcmcf90_amd.workloads import corr_gauss_precision

def rate(kind, d, drs, n, group, upd=0):
    os.environ["MCMCX_GROUP"] = "1" if group else "0"
    its = 200
    ckw = dict(nsimu=100 + its + 1, adaptint=1000, updatesigma=upd, drscale=drs, N0=1.0, S02=0.5)
    pkw = dict(kind=kind, npar=d, par0=np.zeros(d), cmat0=(0.5 / d) * np.eye(d), sigma2=1.0, nobs=11)
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
    return (n * its + (t1["drtries"] - t0["drtries"])) / (ms * 1e-3), ms / its * 1e3, k

print("%-7s %4s %4s %3s %8s | %10s %9s | %10s %9s | %5s" % ("target", "npar", "drs", "s2", "chains", "lane p/s", "us/it", "group p/s", "us/it", "x"))
for kind, d, drs, upd, counts in (("gauss", 40, 0.0, 0, (64, 16384, 131072)), ("gauss", 50, 0.0, 0, (64, 1024, 16384, 131072)), ("gauss", 64, 0.0, 0, (64, 16384, 65536)),
                                  ("banana", 50, 0.0, 0, (64, 131072)),
                                  ("gauss", 2, 2.0, 1, (64, 1024, 16384)), ("gauss", 10, 0.0, 1, (64, 1024, 16384, 65536)), ("gauss", 20, 2.0, 1, (64, 1024, 16384, 65536, 262144)), ("gauss", 20, 0.0, 1, (16384, 262144))):
    for n in counts:
        rl, ul, kl = rate(kind, d, drs, n, False, upd)
        rg, ug, kg = rate(kind, d, drs, n, True, upd)
        print("%-7s %4d %4.1f %3d %8d | %10.3e %9.2f | %10.3e %9.2f | %5.2f  %s / %s" % (kind, d, drs, upd, n, rl, ul, rg, ug, rg / rl, kl, kg), flush=True)
