"""Per-chain SCAM step rate at d = 200 (BASELINE config 5, replicas form): python tools/scam_rate_probe.py NCHAINS ITS"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmcf90_amd import engine_from_problem
from mcmcf90_amd.workloads import problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
its = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ckw, pkw, per_it = problem("c5", 2 + its, adaptint=1000)
e = engine_from_problem(ckw, pkw, nchains=n)
e.init(); e.run(2); e.sync()
t0 = time.perf_counter(); e.run(2 + its); e.sync(); dt = time.perf_counter() - t0
print("per-chain SCAM d=200, %d chains: %.3f s per iteration = %.3e componentwise proposals/s" % (n, dt / its, n * per_it * its / dt))
e.close()
