"""Timing probe for the per-chain SVD tick of method='scam' (adapt_kernel's symsvd_dev), config 5's target:

    python tools/svd_probe.py [d=100] [nchains=32768] [adaptint=50]

prints the time of the iteration that carries the adaptation tick minus the time of a plain iteration, and a checksum
of the rotation (to compare builds: the arithmetic must not change)."""
import os, sys, time, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()
from mcmcf90_amd import engine_from_problem
from mcmcf90_amd.workloads import illcond_gauss_precision as c5_precision

d = int(sys.argv[1]) if len(sys.argv) > 1 else 100
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
adaptint = int(sys.argv[3]) if len(sys.argv) > 3 else 50
ckw = dict(nsimu=adaptint + 2, method="scam", adaptint=adaptint, updatesigma=0, condmax=1e15)
pkw = dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=1e-6 * np.eye(d), mu=np.zeros(d), lam=c5_precision(d))
eng = engine_from_problem(ckw, pkw, nchains=nch)
eng.init(); eng.sync()
t0 = time.perf_counter(); eng.run(adaptint - 1); eng.sync(); t_pre = time.perf_counter() - t0
t0 = time.perf_counter(); eng.run(adaptint); eng.sync(); t_tick = time.perf_counter() - t0
t0 = time.perf_counter(); eng.run(adaptint + 1); eng.sync(); t_it = time.perf_counter() - t0
print("d=%d chains=%d: %d iterations %.2fs; iteration+tick %.2fs; iteration %.3fs; tick %.2fs"
      % (d, nch, adaptint - 2, t_pre, t_tick, t_it, t_tick - t_it))
h = hashlib.sha256()
for c in (0, 1, nch - 1):
    h.update(np.ascontiguousarray(eng.R(c)).tobytes())
print("rotation checksum", h.hexdigest()[:16])
