"""Timing probe for method='scam' on BASELINE config 5's target (not a test, not the bench):

    python tools/scam_probe.py [d=200] [nchains=128] [nsimu=12] [adaptint=5] [pooled=0]

prints init / run time and componentwise proposals per second; in the per-chain mode it also checks chain 1 against
the oracle.  With MCMCX_LIBRARY pointing at a -DMCX_PHASE_PROF build the kernels print
their phase timers."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()
from mcmcf90_amd import engine_from_problem
from oracle import pyoracle as po


from mcmcf90_amd.workloads import illcond_gauss_precision as c5_precision

d = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
nsimu = int(sys.argv[3]) if len(sys.argv) > 3 else 12
adaptint = int(sys.argv[4]) if len(sys.argv) > 4 else 5
pooled = int(sys.argv[5]) if len(sys.argv) > 5 else 0
lam = c5_precision(d)
ckw = dict(nsimu=nsimu, method="scam", adaptint=adaptint, updatesigma=0, condmax=1e15)
pkw = dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=1e-6 * np.eye(d), mu=np.zeros(d), lam=lam)
eng = engine_from_problem(ckw, pkw, nchains=nch, pooled=pooled)
t0 = time.perf_counter(); eng.init(); eng.sync(); print("init %.2fs" % (time.perf_counter() - t0))
t0 = time.perf_counter(); eng.run(); eng.sync(); dt = time.perf_counter() - t0
print("run %.2fs  -> %.3g sub-proposals/s" % (dt, nch * (nsimu - 1) * d / dt), eng.kernel_time())
th = eng.theta()
print("totals", eng.totals())
if not pooled:
    t0 = time.perf_counter()
    o = po.run_chain(po.make_cfg(**ckw), po.Problem(**pkw), chain_id=1)
    print("oracle %.2fs" % (time.perf_counter() - t0))
    print("bit-exact theta chain 1:", np.array_equal(th[1], o.chain[-1, :-1]), "acc", int(o.accepted.sum()))
