"""tools/pooled_occupancy_probe.py -- what two more tiles per CU are worth to pooled_mfma_kernel<false, true> (GPU box).

The tile's LDS vector ((npar rounded up to four) rows of 512 bytes) lets eight waves on a CU up to npar 40 and six at npar 50 (BASELINE config 4).
A variant library built with -DMCX_PROBE_POOLED_LDS_ROWS=52 allocates npar 50's 52 rows whatever npar is: at npar 38 / 40 the same kernel doing the
same work then runs six waves per CU instead of eight.  The ratio of the two libraries at npar 38 / 40 is the prize of any restructuring that
would fit eight tiles at npar 50 (docs/history/r06.md section 10b), before its own costs.  1 048 576 chains, pooled RAM, 300 iterations."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmcf90_amd import engine_from_problem
from mcmcf90_amd.workloads import corr_gauss_precision

for d in (36, 40, 44, 50):
    n = 1048576
    ckw = dict(nsimu=401, method="ram", updatesigma=0, adaptint=100)
    pkw = dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=0.01 * np.eye(d), mu=np.zeros(d), lam=corr_gauss_precision(d))
    e = engine_from_problem(ckw, pkw, nchains=n, pooled=1)
    e.init(); e.run(100); e.sync()
    e.kernel_time(reset=True)
    t0 = time.perf_counter(); e.run(400); e.sync(); dt = time.perf_counter() - t0
    ms, nl, ns = e.kernel_time()
    print("npar %2d  %s  %.3f ms per 100 iterations (kernel %.3f)  %.4g proposals/s  lds rows %d" % (d, e.last_kernel(), dt / 3 * 1e3, ms / max(nl, 1), n * 300 / dt,
          max((d + 3) & ~3, int(os.environ.get("PROBE_ROWS", "0")))), flush=True)
    e.close()
