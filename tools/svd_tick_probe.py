"""Time one MCMC_adapt tick of the per-chain SCAM form of BASELINE config 5 (d = 200): python tools/svd_tick_probe.py NCHAINS
(MCMCX_SVD_LANE=1: the one-lane-per-chain routine; MCMCX_SVD_BLOCK=b: block width of svd_blocked_kernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmcf90_amd import engine_from_problem
from mcmcf90_amd.workloads import problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ckw, pkw, _ = problem("c5", 12, adaptint=10)
ckw = dict(ckw, initcmatn=int(os.environ.get("PROBE_INITCMATN", "200")))      # a full-rank covariance at the tick (cmat0 carries weight): the
# factorisation then converges in its usual ~10-25 sweeps instead of running all 60 on a rank-10 matrix
e = engine_from_problem(ckw, pkw, nchains=n)
e.init()
e.run(9); e.sync()
t0 = time.perf_counter(); e.run(10); e.sync(); t1 = time.perf_counter()
e.run(11); e.sync(); t2 = time.perf_counter()
print("nchains %d  block %s lane %s: iteration 10 + tick %.3f s, iteration 11 %.3f s -> tick %.3f s (x%d for 65536 chains: %.1f s)"
      % (n, os.environ.get("MCMCX_SVD_BLOCK", "default"), os.environ.get("MCMCX_SVD_LANE", "0"), t1 - t0, t2 - t1, (t1 - t0) - (t2 - t1),
         65536 // n, ((t1 - t0) - (t2 - t1)) * 65536 / n))
try:
    import ctypes as C, numpy as np
    from oracle import pyoracle as po
    cm, _, _ = e.chaincov(0)
    G = np.asfortranarray(cm.copy()); V = np.zeros_like(G, order="F"); sv = np.zeros(G.shape[0]); DP = C.POINTER(C.c_double)
    print("  sweeps of the pinned routine on chain 0's covariance:", po.lib().mcxo_symsvd(G.shape[0], G.ctypes.data_as(DP), V.ctypes.data_as(DP), sv.ctypes.data_as(DP)),
          " cond %.2e" % (sv[0] / sv[-1]))
except Exception as ex:
    print("  (sweep count unavailable: %s)" % ex)
e.close()
