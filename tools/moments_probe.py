"""Time the pooled-moments launch (moments_kernel + the tree over tiles) of BASELINE config 4's pooled form:
python tools/moments_probe.py [NCHAINS]   (MCMCX_LIBRARY=tools/_build/libmcmcx_X.so for a variant build).  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmcf90_amd import engine_from_problem
from mcmcf90_amd.workloads import problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
ckw, pkw, _ = problem("c4", 300, adaptint=100)
ckw = dict(ckw, drscale=0.0)
e = engine_from_problem(ckw, pkw, nchains=n, pooled=1)
e.init(); e.run(3); e.sync()
e.allreduce_moments(fetch=False); e.sync()
t0 = time.perf_counter()
for _ in range(20):
    e.allreduce_moments(fetch=False)
e.sync()
t1 = time.perf_counter()
print("%d chains, npar %d: pooled moments %.3f ms per call (kind 0)" % (n, pkw["npar"], (t1 - t0) / 20 * 1e3))
t0 = time.perf_counter(); e.run(103); e.sync(); t1 = time.perf_counter()
print("  100 iterations with one pooled RAM tick: %.2f ms" % ((t1 - t0) * 1e3))
e.close()
