"""tools/ram_group_sweep.py -- method = 'ram': group_ram_kernel (factor in registers, MCMCX_RAM_GROUP=1) against the lane-per-chain RAM kernels
(=0) over the chain count and npar: chain-iterations/s of 200 iterations.  GPU box.  Decides RAM_GROUP_MAX_CHAINS (mcx_api.hip)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmcf90_amd import engine_from_problem
from mcmcf90_amd.workloads import corr_gauss_precision
start = sys.argv[1] if len(sys.argv) > 1 else "default"
for d in (50, 20, 10):
    for n in (64, 1024, 4096, 8192, 16384, 32768, 65536, 131072):
        r = []
        for fam in ("1", "0"):
            os.environ["MCMCX_RAM_GROUP"] = fam
            ckw = dict(nsimu=401, adaptint=100, updatesigma=0, method="ram")
            lam = corr_gauss_precision(d)
            pkw = dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=(np.linalg.inv(lam) if start == "target" else 0.01 * np.eye(d)), mu=np.zeros(d), lam=lam)
            e = engine_from_problem(ckw, pkw, nchains=n)
            e.init(); e.run(201); e.sync()
            t0 = time.perf_counter(); e.run(401); e.sync(); dt = time.perf_counter() - t0
            r.append((n * 200 / dt, e.last_kernel()))
            e.close()
        print("npar %3d %7d chains (%s): group %.3e  lane %.3e  ratio %.2f   [%s | %s]" % (d, n, start, r[0][0], r[1][0], r[0][0] / r[1][0], r[0][1], r[1][1]), flush=True)
