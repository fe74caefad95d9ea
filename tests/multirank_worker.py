"""One rank of a pooled-mode run whose ranks share GPU 0 and exchange through the host transport (tests/test_gpu_multirank.py).

    python tests/multirank_worker.py KEY RANK NRANKS OUTDIR [MODE]

Installs the engine's signal handlers, runs a pooled AM chain and writes what came back (return code, simuind, a digest of
the state) to OUTDIR/rank<R>.json; OUTDIR/rank<R>.ready appears once the run is under way.  MODE:
  long    (default) a very long chain in ONE mcmcx_run call: the test signals one rank from outside
  tail    adaptend = 100: past it no tick lies ahead; the chain runs in calls of 2000 iterations, the test signals one rank
  resume  rank 1 raises SIGUSR1 on itself before the run: both ranks stop at the first tick (applied), clear the flag and run on
  plain   the run of `resume` without a signal"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    key, rank, nranks, outdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    mode = sys.argv[5] if len(sys.argv) > 5 else "long"
    import numpy as np
    from mcmcf90_amd import engine_from_problem, Comm, _lib
    L = _lib.load()
    comm = Comm(key, rank, nranks, 0, backend="host")
    d, n = 4, 256
    ckw = dict(nsimu=2000000, adaptint=50, updatesigma=0)
    if mode == "tail":
        ckw = dict(nsimu=1200000, adaptint=50, updatesigma=0, adaptend=100)
    elif mode in ("resume", "plain"):
        ckw = dict(nsimu=1000, adaptint=50, updatesigma=0)
    pkw = dict(kind="banana", npar=d, par0=np.zeros(d), cmat0=np.eye(d), b=0.1)
    e = engine_from_problem(ckw, pkw, nchains=n, chain_id0=rank * n, pooled=1, comm=comm)
    assert L.mcmcx_install_signal_handlers() == 0
    e.init()
    e.run(100)                                            # two ticks with all ranks: the run is under way
    open(os.path.join(outdir, "rank%d.ready" % rank), "w").write(str(os.getpid()))
    stops = []
    if mode == "tail":                                    # no tick ahead of any of these calls: a signal ends the rank's own run
        rc = 0
        while rc == 0 and e.simuind < e.nsimu:
            rc = e.run(min(e.simuind + 2000, e.nsimu))
    elif mode in ("resume", "plain"):
        import signal
        if mode == "resume" and rank == 1:
            os.kill(os.getpid(), signal.SIGUSR1)          # the flag is up before the call: the ranks agree to stop at its first tick
        rc = e.run()
        while rc == 2:                                    # MCMCX_INTERRUPTED: note where, clear, run on
            stops.append(e.simuind)
            L.mcmcx_clear_interrupt()
            rc = e.run()
    else:
        rc = e.run()
    th = e.theta()
    cm, mean, W, R = e.pooled()
    json.dump({"rc": rc, "simuind": e.simuind, "theta_sum": float(th.sum()), "W": W, "R00": float(R[0, 0]), "stops": stops,
               "theta_bits": th.view(np.uint64).sum(dtype=np.uint64).item(), "R_bits": np.ascontiguousarray(R).view(np.uint64).sum(dtype=np.uint64).item()},
              open(os.path.join(outdir, "rank%d.json" % rank), "w"))
    e.close()
    comm.close()
