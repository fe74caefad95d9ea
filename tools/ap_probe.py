"""tools/ap_probe.py -- wall time of an AP run (adapthist > 1: covmat's batch branch at every adaptation) at config 4's size,
with the blocked batch kernels and with the row-by-row form (MCMCX_COV_BATCH_ROWS=1).  Run on the GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmcf90_amd import engine_from_problem
from mcmcf90_amd.workloads import problem

for rows in (0, 1, 0):
    os.environ["MCMCX_COV_BATCH_ROWS"] = str(rows)
    ckw, pkw, _ = problem("c4", 801, adaptint=100)
    ckw = dict(ckw, method="dram", drscale=0.0, adapthist=200)
    e = engine_from_problem(ckw, pkw, nchains=131072, chain_id0=0)
    e.init(); e.run(301); e.sync()
    t0 = time.perf_counter(); e.run(801); e.sync(); dt = time.perf_counter() - t0
    print("batch branch %s: 500 iterations with 5 AP adaptations of 131072 chains, d = 50: %.1f ms (%.3g proposals/s)"
          % ("row by row" if rows else "in blocks", dt * 1e3, 131072 * 500 / dt))
    e.close()
