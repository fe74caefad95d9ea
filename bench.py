#!/usr/bin/env python3
"""bench.py -- MH proposals/sec of the MI355X engine on BASELINE.json's headline workload.

Workload "C4": correlated Gaussian target d=50 (Sigma_ij = 0.5^|i-j|, Lambda dense), method='ram'
(MCMC_run_ram + MCMC_adapt_ram with a per-chain Cholesky factor), 131072 chains per GPU (the 8-GPU
configuration of BASELINE.json is 1 048 576 chains = 131072 x 8: weak scaling), pooled
empirical-moment reduction of all chains every `--its-per-step` iterations (all-reduce over RCCL
when N > 1).  One bench "step" = --its-per-step MH iterations of every chain + that reduction.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --workload c5              # the other BASELINE configurations: c2, c3, c5 (see WORKLOADS)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events on the engine's
stream around every step-kernel launch of the timed region (mcmcx_kernel_time); `cpu_baseline`
times the real Fortran reference (oracle/_ref, kind "reference") or the C oracle (kind "port")
on one host core on the same target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
FP64_MFMA_PEAK_TF = 78.6       # MI355X FP64 matrix (= FP64 vector) spec peak; tools/mfma_f64_probe.hip measures 77.6

WORKLOADS = {   # BASELINE.json configs 2-5; the headline metric is quoted on c4
    "c2": "C2: isotropic Gaussian d=10, AM (method=dram, drscale=0)",
    "c3": "C3: banana d=20, DRAM (2-stage delayed rejection, drscale=2)",
    "c4": "C4: correlated Gaussian d=50 (Sigma=0.5^|i-j|)",
    "c5": "C5: ill-conditioned Gaussian d=200 (cond 1e6), SCAM componentwise",
}
DEFAULT_CHAINS = {"c2": 65536, "c3": 262144, "c4": 131072, "c5": 65536}


def alg_bytes_per_proposal(d, method):
    """Algorithmic HBM bytes per proposal:
    theta read + write (16 d) + ss/prior read/write (32); per-chain Cholesky factor, packed
    upper triangle: RAM reads it once and writes it once per iteration (8 d (d+1)), AM / DRAM only read
    it (4 d (d+1)); per-chain SCAM streams its rotation twice per componentwise proposal (16 d^2)."""
    base = 16 * d + 32
    tri = d * (d + 1) // 2 * 8
    if method == "pooled":
        return base                                   # the shared factor lives in the scalar cache
    if method == "scam":
        return base + 16 * d * d
    return base + (2 * tri if method == "ram" else tri)


def cpu_baseline(wl, ckw, pkw, per_it, label, target_seconds=12.0):
    """One chain of the same workload on one host core (the reference adapts its single chain on its own history)."""
    from oracle import pyoracle as po, refrun as rr
    prob = po.Problem(**pkw)
    out = {"cores": 1, "unit": "proposals/s"}
    # the C oracle, in process (no file output): a short run sizes the reference run
    n0 = 2000 if per_it == 1 else 12
    t0 = time.perf_counter(); po.run_chain(po.make_cfg(**dict(ckw, nsimu=n0)), prob); t_probe = time.perf_counter() - t0
    n_port = int(min(200000, max(n0, n0 * 4.0 / max(t_probe, 1e-3))))
    cfgp = po.make_cfg(**dict(ckw, nsimu=n_port))
    t0 = time.perf_counter(); o = po.run_chain(cfgp, prob); t_port = time.perf_counter() - t0
    port_rate = ((n_port - 1) * per_it + o.drtries) / t_port
    if rr.available():
        try:
            nsimu = int(min(1500000, max(n0, port_rate / per_it * target_seconds * 0.6)))
            cfg = po.make_cfg(**dict(ckw, nsimu=nsimu))
            t0 = time.perf_counter(); r = rr.run_reference(cfg, prob, timeout=300, pinned_svd=bool(cfg.usesvd)); t_ref = time.perf_counter() - t0
            o2 = po.run_chain(cfg, prob)             # same stream: its delayed-rejection count is the reference's
            out["all_cores"] = cpu_all_cores(wl, ckw, per_it, port_rate)
            out.update(value=((nsimu - 1) * per_it + o2.drtries) / t_ref, kind="reference",
                       sample="mcmcf90 Fortran reference (flang -O2 + MKL, oracle/_ref/mcxref), 1 chain, %s, "
                              "nsimu=%d, wall time of the whole program incl. namelist/file I/O = %.2f s" % (label, nsimu, t_ref),
                       port_value=port_rate)
            return out
        except Exception as ex:                      # reference binary present but not runnable here
            out["reference_error"] = str(ex)[:200]
    out["all_cores"] = cpu_all_cores(wl, ckw, per_it, port_rate)
    out.update(value=port_rate, kind="port",
               sample="C oracle (oracle/mcx_oracle.c, gcc -O2), 1 chain, %s, nsimu=%d, %.2f s" % (label, n_port, t_port))
    return out


def cpu_all_cores(wl, ckw, per_it, port_rate, seconds=4.0):
    """every host core runs one independent chain of the C restatement, each in a process of its
    own (the reference is one chain per process; the port is what spreads over the cores without N copies of its file I/O)."""
    import subprocess
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    avail = cores
    cores = min(cores, 32)                            # bounded: the default bench run must stay within minutes
    n = int(max(50, port_rate / per_it * seconds))
    cmd = [sys.executable, "-m", "oracle.portrun", wl, str(n), str(ckw.get("adaptint", 100))]
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd + [str(1000 + c), ckw.get("method", "dram")], cwd=ROOT, stdout=subprocess.PIPE) for c in range(cores)]
    tries = 0
    for p in procs:
        try:
            out, _ = p.communicate(timeout=90)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            return {"error": "the host did not finish %d port chains in 90 s (CPU quota?)" % cores}
        if p.returncode != 0:
            return {"error": "oracle.portrun failed"}
        tries += int(out.decode().split()[-1])
    dt = time.perf_counter() - t0
    return {"value": (cores * (n - 1) * per_it + tries) / dt, "unit": "proposals/s", "cores": cores, "kind": "port",
            "sample": "C oracle, %d independent chains, one process each (host CPUs in the affinity mask: %d, processes capped at 32), nsimu=%d each, %.2f s incl. process start" % (cores, avail, n, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS), help="BASELINE.json configuration (default: the headline one)")
    ap.add_argument("--chains-per-gpu", type=int, default=0, help="default: the configuration's chain count (c4: 131072 = 1048576/8)")
    ap.add_argument("--its-per-step", type=int, default=0, help="MH iterations per bench step (default 100; c5: 10)")
    ap.add_argument("--method", default=None, choices=["ram", "dram"], help="c4 only: per-chain RAM (default) or AM")
    ap.add_argument("--pooled", action="store_true", help="one shared factor from the all-reduced pooled covariance (c5: the default)")
    ap.add_argument("--replicas", action="store_true", help="c5: per-chain rotations (the reference's semantics) instead of the pooled one")
    ap.add_argument("--start", default="default", choices=["default", "target"],
                    help="c4: 'target' starts from cmat0 = Sigma, i.e. at RAM's target acceptance rate, where most iterations are "
                         "Cholesky downdates (default: cmat0 = 0.01 I, 86 %% accepted, RAM adapts by updates)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--one-gpu-dryrun", action="store_true",
                    help="debug: all ranks share GPU 0 and reduce over gloo (checks the N>1 control path on a 1-GPU box)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    if a.one_gpu_dryrun:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.one_gpu_dryrun:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def all_reduce_dev(t, op=None):
        """Sum a device tensor over ranks: RCCL in place, or (dry run) through a host copy over gloo."""
        kw = {} if op is None else {"op": op}
        if a.one_gpu_dryrun:
            h = t.cpu(); dist.all_reduce(h, **kw); t.copy_(h)
        else:
            dist.all_reduce(t, **kw)

    from mcmcf90_amd import engine_from_problem
    from mcmcf90_amd.workloads import problem
    wl = a.workload
    n_local = a.chains_per_gpu or DEFAULT_CHAINS[wl]
    ips = a.its_per_step or (10 if wl == "c5" else 100)       # c5: one iteration is d = 200 componentwise proposals
    nsimu = 1 + (a.warmup + a.steps) * ips
    ckw, pkw, per_it = problem(wl, nsimu, adaptint=max(ips, 100))
    d = pkw["npar"]
    if wl == "c4" and a.start == "target":
        pkw = dict(pkw, cmat0=np.linalg.inv(np.asarray(pkw["lam"], dtype=float)))
    if wl == "c4" and a.method == "dram":
        ckw = dict(ckw, method="dram")
    if wl == "c5":
        a.pooled = not a.replicas
    elif a.pooled:
        ckw = dict(ckw, method="dram", drscale=0.0)
    method = ckw.get("method", "dram")
    eng = engine_from_problem(ckw, pkw, nchains=n_local, chain_id0=rank * n_local, device=local_rank,
                              pooled=1 if a.pooled else 0)
    mom_len = 1 + d + d * (d + 1) // 2
    pooled = torch.zeros(mom_len, dtype=torch.float64, device=dev)
    xbuf = torch.zeros(mom_len, dtype=torch.float64, device=dev)
    if a.pooled and world > 1:
        def _xchg():                                   # called by the engine at every adaptation tick
            all_reduce_dev(xbuf)
            torch.cuda.synchronize()
        eng.set_exchange(_xchg, xbuf.data_ptr())
    eng.init()

    def one_step(k):
        eng.run(1 + (k + 1) * ips)
        eng.pooled_moments_dev(pooled.data_ptr())      # fixed-tree sum over this GPU's chains, stays in HBM
        if world > 1:
            eng.sync()                                 # engine stream -> torch stream hand-off
            all_reduce_dev(pooled)                     # RCCL over xGMI: 1+d+d(d+1)/2 doubles

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(a.warmup):
        one_step(k)
    fence()
    eng.kernel_time(reset=True)
    t0 = time.perf_counter()
    for k in range(a.warmup, a.warmup + a.steps):
        one_step(k)
    fence()
    dt = time.perf_counter() - t0
    kms, klaunch, ksteps = eng.kernel_time()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        all_reduce_dev(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    tot = eng.totals()
    tries = torch.tensor([float(tot["drtries"])], dtype=torch.float64, device=dev)   # delayed-rejection proposals of this rank (whole run)
    if world > 1:
        all_reduce_dev(tries)
    if rank == 0:
        base = float(world) * n_local * ips * per_it
        dr_per_it = float(tries.item()) / (a.warmup + a.steps) / ips                # stage-2 proposals per iteration, all ranks
        proposals = (base + dr_per_it * ips) * a.steps
        value = proposals / dt
        per_launch_prop = proposals / world / max(klaunch, 1)                        # proposals one launch of one GPU evaluates
        avg_launch_s = kms / 1e3 / max(klaunch, 1)
        if wl == "c5" and a.pooled:                   # shared rotation: three d x d products per proposal on the f64 matrix cores
            flop = 6.0 * d * d
            achieved = flop * per_launch_prop / avg_launch_s / 1e12
            roof = {"bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                    "frac": achieved / FP64_MFMA_PEAK_TF, "traffic": None, "kernel": "mcx::scam_pooled_kernel",
                    "alg_flop_per_proposal": flop}
        else:
            balg = alg_bytes_per_proposal(d, "pooled" if a.pooled else method)
            achieved = balg * per_launch_prop / avg_launch_s / 1e9
            traffic = None
            tfile = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tfile):
                try:
                    tj = json.load(open(tfile))
                    key = "%s_d%d" % (method, d)
                    if key in tj and not a.pooled:   # measured HBM bytes per proposal (rocprofv3 --pmc, see profiles/)
                        traffic = tj[key]["hbm_bytes_per_proposal"] * per_launch_prop
                except Exception:
                    traffic = None
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "kernel": "mcx::scam_kernel" if method == "scam" else "mcx::step_kernel", "alg_bytes_per_proposal": balg}
        roof.update(launches=int(klaunch), avg_launch_ms=avg_launch_s * 1e3, kernel_share_of_wall=kms / 1e3 / dt)
        if roof["bound"] == "hbm" and (a.pooled or d <= 20):
            roof["note"] = ("the chip's HBM roof is quoted for uniformity; this configuration is bound by the per-chain random "
                            "numbers (Philox + polar + pinned log/sqrt on the VALU), see DESIGN.md section 5")
        if method == "ram":
            stay = float(tot["stayed"]) / (float(n_local) * (nsimu - 1))
            roof["note"] = ("accepted fraction on this rank %.2f (start: %s); RAM updates the factor after alpha >= alphatarget (one "
                            "read+write sweep) and downdates it otherwise (two sweeps): DESIGN.md section 10 item 6" % (1.0 - stay, a.start))
        mode = method + (" pooled (one shared factor)" if a.pooled else ", per-chain factor")
        cnt = float(pooled[0].item())
        mean = (pooled[1:1 + d] / cnt).cpu().numpy()
        line = {
            "metric": "MH proposals/sec (whole node), d=50 Gaussian target" if wl == "c4" else "MH proposals/sec (whole node), " + WORKLOADS[wl],
            "value": value, "unit": "proposals/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s, method=%s, %d chains/GPU, pooled moments all-reduce every %d iterations"
                                   % (WORKLOADS[wl], mode, n_local, ips),
                       "chains_per_gpu": n_local, "its_per_step": ips, "npar": d, "method": method,
                       "proposals_per_iteration": per_it + dr_per_it / (float(world) * n_local),
                       "parallelism": "chains sharded over %d GPU(s)" % world},
            "roofline": roof,
            "pooled_check": {"chains": cnt, "max_abs_mean": float(np.max(np.abs(mean)))},
        }
        if not a.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(wl, ckw, pkw, per_it, "d=%d %s" % (d, method))
        print(json.dumps(line), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
