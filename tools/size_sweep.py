"""tools/size_sweep.py [am|ram|scam] -- proposals/s over npar at a chain count that fills the chip (131072 chains up to npar 64,
fewer above): kernel chosen, rate, and the factor stream it amounts to (AM reads the packed factor once per proposal, RAM reads
and writes it).  A size at which a kernel choice goes wrong shows up as a step in the last column.  Run on the GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmcf90_amd import engine_from_problem
from mcmcf90_amd.workloads import corr_gauss_precision

what = sys.argv[1] if len(sys.argv) > 1 else "am"
dims = [int(x) for x in os.environ.get("SIZES", "4,8,10,11,12,16,20,24,32,40,50,64,80,100,128,160,200,256").split(",")]
for d in dims:
    n = int(os.environ.get("CHAINS", "0")) or (131072 if d <= 64 else (65536 if d <= 128 else 16384))
    if what == "scam":
        n = 65536 if d <= 32 else 8192
    nit = 401 if what != "scam" else 41
    n0 = 201 if what != "scam" else 21
    ckw = dict(nsimu=nit, adaptint=100 if what != "scam" else 20, updatesigma=0, drscale=0.0)
    if what == "ram":
        ckw["method"] = "ram"
    if what == "scam":
        ckw["method"] = "scam"
    pkw = dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=(0.5 / d) * np.eye(d), mu=np.zeros(d), lam=corr_gauss_precision(d))
    e = engine_from_problem(ckw, pkw, nchains=n, chain_id0=0)
    e.init(); e.run(n0); e.sync()
    t0 = time.perf_counter(); e.run(nit); e.sync(); dt = time.perf_counter() - t0
    per_it = d if what == "scam" else 1
    rate = n * (nit - n0) * per_it / dt
    P = d * (d + 1) // 2
    fac = (2 * d * d * 8) if what == "scam" else (P * 8 * (2 if what == "ram" else 1))
    print("%-5s npar %3d %7d chains  %-34s %9.3g proposals/s  factor stream %6.2f TB/s" % (what, d, n, e.last_kernel()[:34], rate, rate * fac / 1e12), flush=True)
    e.close()
