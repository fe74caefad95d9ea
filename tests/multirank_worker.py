"""One rank of a pooled-mode run whose ranks share GPU 0 and exchange through the host transport (tests/test_gpu_multirank.py).

    python tests/multirank_worker.py KEY RANK NRANKS OUTDIR

Installs the engine's signal handlers, runs a long pooled AM chain in ONE mcmcx_run call and writes what came back
(return code, simuind, a digest of the state) to OUTDIR/rank<R>.json; OUTDIR/rank<R>.ready appears once the run is under way."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    key, rank, nranks, outdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import numpy as np
    from mcmcf90_amd import engine_from_problem, Comm, _lib
    L = _lib.load()
    comm = Comm(key, rank, nranks, 0, backend="host")
    d, n = 4, 256
    ckw = dict(nsimu=2000000, adaptint=50, updatesigma=0)
    pkw = dict(kind="banana", npar=d, par0=np.zeros(d), cmat0=np.eye(d), b=0.1)
    e = engine_from_problem(ckw, pkw, nchains=n, chain_id0=rank * n, pooled=1, comm=comm)
    assert L.mcmcx_install_signal_handlers() == 0
    e.init()
    e.run(100)                                            # two ticks with all ranks: the run is under way
    open(os.path.join(outdir, "rank%d.ready" % rank), "w").write(str(os.getpid()))
    rc = e.run()
    th = e.theta()
    cm, mean, W, R = e.pooled()
    json.dump({"rc": rc, "simuind": e.simuind, "theta_sum": float(th.sum()), "W": W, "R00": float(R[0, 0])},
              open(os.path.join(outdir, "rank%d.json" % rank), "w"))
    e.close()
    comm.close()
