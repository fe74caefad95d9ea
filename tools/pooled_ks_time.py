"""tools/pooled_ks_time.py -- pooled_mfma_ks_kernel (MCMCX_POOLED_KS=1) and pooled_mfma_kernel<false, true> (=0) timed at BASELINE config 4's size; with a
-DMCX_PHASE_PROF build (tools/build_variant.sh phase -DMCX_PHASE_PROF; MCMCX_LIBRARY=...) both print where a wave's iteration goes.  GPU box."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmcf90_amd import engine_from_problem
from mcmcf90_amd.workloads import corr_gauss_precision
for ks in ("1", "0"):
    d, n = 50, 1048576
    os.environ["MCMCX_POOLED_WAVES"] = "2"; os.environ["MCMCX_POOLED_KS"] = ks
    ckw = dict(nsimu=401, method="ram", updatesigma=0, adaptint=100)
    pkw = dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=0.01 * np.eye(d), mu=np.zeros(d), lam=corr_gauss_precision(d))
    e = engine_from_problem(ckw, pkw, nchains=n, pooled=1)
    e.init(); e.run(100); e.sync(); e.kernel_time(reset=True)
    t0 = time.perf_counter(); e.run(400); e.sync(); dt = time.perf_counter() - t0
    ms, nl, ns = e.kernel_time()
    print("%-34s kernel %.3f ms per 100 iterations  %.4g proposals/s" % (e.last_kernel(), ms / max(nl, 1), n * 300 / dt), flush=True)
    e.close()
