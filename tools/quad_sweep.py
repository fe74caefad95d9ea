"""tools/quad_sweep.py -- npar <= 16 with the chip full: lane-per-chain kernels / lane-group kernel with sixteen lanes per chain / with
four (quads): proposals/s of 200 iterations between two adaptations.  GPU box."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmcf90_amd import engine_from_problem
from mcmcf90_amd.workloads import corr_gauss_precision

def rate(kind, d, drs, n, mode, upd=0):
    os.environ["MCMCX_GROUP"] = "0" if mode == "lane" else "1"
    os.environ["MCMCX_GROUP_GW"] = "4" if mode == "quad" else "16"
    its = 200
    ckw = dict(nsimu=100 + its + 1, adaptint=1000, updatesigma=upd, drscale=drs)
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
    e.close()
    return (n * its + (t1["drtries"] - t0["drtries"])) / (ms * 1e-3)

print("%-7s %4s %4s %8s | %10s %10s %10s | quad/lane quad/g16" % ("target", "npar", "drs", "chains", "lane", "group16", "quad"))
for kind in ("gauss", "banana"):
    for d in (2, 4, 8, 10, 12, 16):
        for drs in (0.0, 2.0, 3.0):
            for n in (32768, 65536, 262144):
                r = [rate(kind, d, drs, n, m) for m in ("lane", "g16", "quad")]
                print("%-7s %4d %4.1f %8d | %10.3e %10.3e %10.3e | %6.2f %6.2f" % (kind, d, drs, n, r[0], r[1], r[2], r[2] / r[0], r[2] / r[1]), flush=True)
