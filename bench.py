#!/usr/bin/env python3
"""bench.py -- MH proposals/sec of the MI355X engine on BASELINE.json's headline workload.

Workload "C4": correlated Gaussian target d=50 (Sigma_ij = 0.5^|i-j|, Lambda dense), method='ram'
(MCMC_run_ram + MCMC_adapt_ram with a per-chain Cholesky factor), BASELINE.json's 1 048 576 chains divided among
the --gpus of the run (strong scaling: all of them on the one GPU at N = 1, 131072 per GPU at N = 8; `--plan` prints
the sharding of any N without touching a GPU; rounds 1-3 ran 131072 chains per GPU at every N -- the line's
`config.total_chains` says which problem a number belongs to), pooled
empirical-moment reduction over all chains of all GPUs every `--its-per-step` iterations (RCCL
all-gather + fixed tree inside libmcmcx.so when N > 1).  One bench "step" = --its-per-step MH
iterations of every chain + that reduction.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8                   # starts 8 ranks itself, one process per GPU
    python bench.py --workload c5              # the other BASELINE configurations: c2, c3, c5 (see WORKLOADS)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W      # the same ranks under torchrun

One process per GPU.  Under torchrun the ranks come from RANK / LOCAL_RANK / WORLD_SIZE; without it
`--gpus N` spawns the N rank processes itself (before anything touches a GPU) and fails if they cannot
all start.  The ranks meet in libmcmcx.so's communicator (RCCL; the ncclUniqueId travels through a POSIX
shm segment), which also carries the barrier and the max-over-ranks of the timing: torch is not imported.
Every rank prints its device (name, PCI bus id) and its place in the communicator to stderr once, runs RCCL
with NCCL_DEBUG=WARN, and arms a watchdog around the communicator's formation (--comm-timeout): a rank that
cannot join ends the run with a non-zero exit code instead of hanging it.  If RCCL itself refuses to form the run ends non-zero with RCCL's
own diagnostics and NO line -- unless `--allow-host-transport` was given: then (every rank having its own GPU) the same latency-sized exchange
goes through the shared-memory segment and the line says so: `transport` "host", `rccl_error`, `rccl_ranks` 0.

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events on the engine's
stream around every step-kernel launch of the timed region (mcmcx_kernel_time); `cpu_baseline`
times the real Fortran reference (oracle/_ref, kind "reference") or the C oracle (kind "port")
on one host core and on all of them on the same target; `other_configs` (the LAST key, compact entries of <= 160 characters:
value, frac, bound, issue, kernel, ms, share -- `--verbose` or gpurun_out/bench_other_configs.json for the long form) holds, at N = 1,
a short run of each of the other BASELINE configurations in the same process after the headline's timed region and, at N > 1, a short
run of c4 in POOLED mode on the same communicator -- the one collective that sits on the critical path (the pooled RAM tick every
adaptint iterations) -- so that one scaling run also measures that; `roofline.others` mirrors [value, frac] of every BASELINE
configuration; `roofline.traffic_source` says where the (stored, not live) PMC figure comes from; `device` identifies the box.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import threading
import time
import uuid

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
FP64_MFMA_PEAK_TF = 78.6       # MI355X FP64 matrix (= FP64 vector) spec peak; tools/mfma_f64_probe.hip measures 77.6
# vector-instruction issue roof: 256 CUs x 4 SIMDs, one wave64 VALU instruction per SIMD every 4 cycles at 2.4 GHz
# (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost': v_fma 4 cyc for one wave's stream on one SIMD)
VALU_ISSUE_PEAK_GIPS = 256 * 4 * 2.4 / 4.0 * 64 / 64      # 614.4 G wave-instructions/s

WORKLOADS = {   # BASELINE.json configs 2-5; the headline metric is quoted on c4
    "c1": "C1 at scale: config 1's decay model (npar 2, device-resident), DRAM drscale=2, updatesigma=1",
    "c1x": "C1 with response columns: nycol=2 (one sigma2 per column), npar 3, DRAM drscale=2, updatesigma=1",
    "c2": "C2: isotropic Gaussian d=10, AM (method=dram, drscale=0)",
    "c3": "C3: banana d=20, DRAM (2-stage delayed rejection, drscale=2)",
    "c4": "C4: correlated Gaussian d=50 (Sigma=0.5^|i-j|)",
    "c5": "C5: ill-conditioned Gaussian d=200 (cond 1e6), SCAM componentwise",
}
DEFAULT_CHAINS = {"c1": 262144, "c1x": 262144, "c2": 65536, "c3": 262144, "c4": 1048576, "c5": 65536}
# BASELINE.json: c4 and c5 are ONE problem "sharded over 8 x MI355X" -- the whole job's chains are divided among the GPUs of a run
# (strong scaling); c2 and c3 are quoted "on 1 MI355X" and keep their count per GPU (weak scaling)
STRONG = ("c4", "c5")


def default_chains(wl, world):
    if wl in STRONG:
        n = DEFAULT_CHAINS[wl] // world
        return max(64, n - n % 64)
    return DEFAULT_CHAINS[wl]


def shard_plan(wl, world, chains_per_gpu=0):
    """What `--gpus world` runs: rank r owns the chains [r n, (r + 1) n) (chain_id0 = r n keys their Philox streams, so the chains are the
    same whatever the GPU count) -- n the configuration's chains divided by world for c4 / c5 (strong scaling: BASELINE.json
    quotes them as ONE problem sharded over the node), the configuration's count per GPU for c2 / c3 (weak), or --chains-per-gpu."""
    n = chains_per_gpu or default_chains(wl, world)
    return {"workload": wl, "n_gpus": world, "chains_per_gpu": n, "total_chains": n * world,
            "scaling": "strong" if (wl in STRONG and not chains_per_gpu) else "weak",
            "ranks": [{"rank": r, "chain_id0": r * n, "chains": n, "device": r} for r in range(world)],
            "collective": "none on the data path; pooled moments: all-gather of %s doubles per rank + fixed pairwise tree over ranks" % "1 + d + d (d + 1) / 2"}


def alg_bytes_per_proposal(d, method, down_frac=0.0):
    """Algorithmic HBM bytes per proposal:
    theta read + write (16 d) + ss/prior read/write (32); per-chain Cholesky factor, packed
    upper triangle: RAM reads it once and writes it once per iteration (8 d (d+1)), AM / DRAM only read
    it (4 d (d+1)); per-chain SCAM streams its rotation twice per componentwise proposal (16 d^2).
    A RAM downdate (dchdd.f:141-179) is two dependent sweeps -- the forward substitution, then rotations generated
    from its LAST element backwards -- so it reads the factor twice: `down_frac` of the proposals add one read."""
    base = 16 * d + 32
    tri = d * (d + 1) // 2 * 8
    if method == "pooled":
        return base                                   # the shared factor lives in the scalar cache
    if method == "scam":
        return base + 16 * d * d
    if method == "ram":
        return base + 2 * tri + int(round(down_frac * tri))
    return base + tri


def kernels_sha():
    from mcmcf90_amd.build import source_sha
    return source_sha()


def ab_switches():
    """Environment switches that select another kernel path or another library build than the profiled one
    (mcx_api.hip's A/B switches, tools/build_variant.sh): with any of them set the stored PMC figures do not describe
    what is being timed."""
    return sorted(k for k in os.environ if k.startswith("MCMCX_") and k not in ("MCMCX_COMM_KEY",))


def measured_counters(key):
    """PMC results of profiles/traffic.json for this configuration -- only when they were collected with the engine
    sources that are running now (the file records the sha of csrc/*: kernels, device functions, host orchestration and
    communicator) and no A/B switch is active."""
    if ab_switches():
        return None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tj = json.load(open(tfile))
        e = tj.get(key)
        if e and e.get("kernels_sha") == kernels_sha():
            return e
    except Exception:
        pass
    return None


# ------------------------------------------------------------------ the CPU baseline (rank 0, N = 1): reference + port
def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cgroup_cpu_quota():
    """CPUs the container may use at once (cgroup v2 cpu.max / v1 cpu.cfs_quota_us), or None when unlimited / unknown."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except Exception:
        return None


def reference_all_cores(ckw, pkw, per_it, rate_it, cores):
    """SURVEY section 8(d)(ii) / north_star: the REFERENCE on the node's own host cores -- one process of oracle/_ref/mcxref
    per CPU of the affinity mask, each an independent chain (its own Philox stream) in its own tmpfs directory.  Sampling
    loop only, by the same two-nsimu difference as the one-core figure: all processes run nsimu = n1 together, then all run
    nsimu = n2; rate = cores x (proposals(n2) - proposals(n1)) / (wall(n2) - wall(n1))."""
    from oracle import pyoracle as po, refrun as rr
    prob = po.Problem(**pkw)
    d = prob.npar
    cap = int(48e6 / (16.0 * (d + 2)))                  # chain array + its .mat file: <= ~48 MB per process
    n2 = int(max(16, min(cap, rate_it * 1.5)))
    n1 = max(8, n2 // 4)
    def batch(n):
        cfg = po.make_cfg(**dict(ckw, nsimu=n))
        r = rr.run_reference_many(cfg, prob, nprocs=cores, chain_id0=1000, timeout=240, pinned_svd=bool(cfg.usesvd))
        tries = 0
        if cfg.dodr:                                    # same streams: the port's delayed-rejection count is the reference's
            tries = sum(po.run_chain(cfg, prob, chain_id=1000 + c).drtries for c in range(min(cores, 4))) / float(min(cores, 4)) * cores
        return cores * (n - 1) * per_it + tries, r

    batch(n1)                                           # untimed: binary, MKL, tmpfs paged in and every (virtual) core awake
    res = [batch(n1), batch(n2)]
    again = batch(n1)                                   # the short batch once more: the less disturbed of the two counts
    if again[1]["wall"] < res[0][1]["wall"]:
        res[0] = again
    (p1, r1), (p2, r2) = res
    dt = r2["wall"] - r1["wall"]
    if not dt > 0:
        raise RuntimeError("reference all-cores: wall(n2) <= wall(n1) (%.2f, %.2f)" % (r2["wall"], r1["wall"]))
    quota = cgroup_cpu_quota()
    return {"value": (p2 - p1) / dt, "unit": "proposals/s", "cores": cores, "kind": "reference",
            "effective_parallelism": r2["cpu_seconds"] / r2["wall"],
            "cgroup_cpu_quota": quota,
            "sample": "reference (oracle/_ref/mcxref), %d processes x 1 chain on %s: loop only = %d x (prop(nsimu=%d) - prop(nsimu=%d)) / (%.2f - %.2f s); "
                      "CPU s / wall = %.1f" % (cores, r2["scratch"], cores, n2, n1, r2["wall"], r1["wall"], r2["cpu_seconds"] / r2["wall"])}


def port_all_cores(wl, ckw, per_it, port_rate, cores, seconds=0.6):
    """The C restatement, one process per CPU of the affinity mask (kept beside the reference figure)."""
    n = int(max(50, port_rate / per_it * seconds))
    cmd = [sys.executable, "-m", "oracle.portrun", wl, str(n), str(ckw.get("adaptint", 100))]
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd + [str(1000 + c), ckw.get("method", "dram")], cwd=ROOT, stdout=subprocess.PIPE) for c in range(cores)]
    tries = 0
    for p in procs:
        try:
            out, _ = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            return {"error": "the host did not finish %d port chains in 120 s (CPU quota?)" % cores}
        if p.returncode != 0:
            return {"error": "oracle.portrun failed"}
        tries += int(out.decode().split()[-1])
    dt = time.perf_counter() - t0
    return {"value": (cores * (n - 1) * per_it + tries) / dt, "unit": "proposals/s", "cores": cores, "kind": "port",
            "sample": "C oracle, %d processes x 1 chain, nsimu=%d each, %.2f s incl. process start" % (cores, n, dt)}


def cpu_baseline(wl, ckw, pkw, per_it, label, target_seconds=12.0, all_cores=True):
    """One chain of the same workload on one host core (the reference adapts its single chain on its own history), then
    one chain per host core."""
    from oracle import pyoracle as po, refrun as rr
    prob = po.Problem(**pkw)
    cores = host_cores()
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
            # the sampling loop alone: two runs of the reference program, nsimu = n and n/4, in a tmpfs scratch
            # directory; the difference removes process start, namelist / input files and MCMC_init
            n2 = int(min(1000000, max(4 * n0, port_rate / per_it * target_seconds * 0.5)))
            n1 = max(n0, n2 // 4)
            runs = []
            rr.run_reference(po.make_cfg(**dict(ckw, nsimu=n0)), prob, timeout=300, pinned_svd=bool(cfgp.usesvd), timing_only=True)   # page the binary + MKL in
            for n in (n1, n2):
                cfg = po.make_cfg(**dict(ckw, nsimu=n))
                r = rr.run_reference(cfg, prob, timeout=300, pinned_svd=bool(cfg.usesvd), timing_only=True)
                o2 = po.run_chain(cfg, prob)             # same stream: its delayed-rejection count is the reference's
                runs.append(((n - 1) * per_it + o2.drtries, r.seconds, r.scratch))
            (p1, t1, _), (p2, t2, scratch) = runs
            value = (p2 - p1) / (t2 - t1)
            out.update(value=value, kind="reference",
                       sample="mcmcf90 Fortran reference (flang -O2 + MKL, oracle/_ref/mcxref), 1 chain, %s: loop only = "
                              "(prop(nsimu=%d) - prop(nsimu=%d)) / (%.2f - %.2f s), outputs on %s" % (label, n2, n1, t2, t1, scratch),
                       whole_program_value=p2 / t2, port_value=port_rate)
            if all_cores:
                try:
                    out["all_cores"] = reference_all_cores(ckw, pkw, per_it, value / (per_it + o2.drtries / max(n2 - 1.0, 1.0)), cores)
                except Exception as ex:
                    out["all_cores"] = {"error": str(ex)[:300], "kind": "reference", "cores": cores}
                out["all_cores_port"] = port_all_cores(wl, ckw, per_it, port_rate, cores)
            return out
        except Exception as ex:                      # reference binary present but not runnable here
            out["reference_error"] = str(ex)[:200]
    if not rr.available():
        # the evidence pipeline lost its reference (oracle/_ref is a build product: a clean clone on a GPU box has none) -- said in the
        # line and on stderr, not only through `kind`
        out["reference_missing"] = True
        sys.stderr.write("bench.py: oracle/_ref/mcxref (the reference compiled from /root/reference) is ABSENT: cpu_baseline is the C port (kind 'port')\n")
    if all_cores:
        out["all_cores"] = port_all_cores(wl, ckw, per_it, port_rate, cores, seconds=3.0)
    out.update(value=port_rate, kind="port",
               sample="C oracle (oracle/mcx_oracle.c, gcc -O2), 1 chain, %s, nsimu=%d, %.2f s" % (label, n_port, t_port))
    return out


def device_ident(L, dev):
    """What tells this box from another one of the pool in the JSON line: UUID, PCI bus id, clock limits (HIP) and, where
    rocm-smi answers, the clocks and the power cap it reports right now (best effort; never fatal)."""
    import ctypes
    out = {}
    try:
        uuid_hex = ctypes.create_string_buffer(40)
        v = (ctypes.c_int32 * 5)()
        if L.mcmcx_device_ident(dev, uuid_hex, 40, v) == 0:
            out.update(uuid=uuid_hex.value.decode(), sclk_limit_mhz=v[0] / 1e3, mclk_limit_mhz=v[1] / 1e3, mem_bus_bits=int(v[2]), cus=int(v[3]), l2_bytes=int(v[4]))
        info = ctypes.create_string_buffer(256)
        L.mcmcx_device_info(dev, info, 256)
        txt = info.value.decode(errors="replace")
        out["info"] = txt
        if "pci " in txt:
            out["pci"] = txt.split("pci ", 1)[1].split(",")[0]
    except Exception as ex:
        out["error"] = str(ex)[:120]
    try:
        # (under rocprofv3 the profiler's preloaded library has initialised the GPU in every child too, and rocm-smi is a `#!/usr/bin/env python3`
        #  script: an exec after GPU initialisation, which the GPU boxes refuse -- skip it there)
        if "rocprof" in os.environ.get("LD_PRELOAD", "") or os.environ.get("ROCPROFILER_REGISTER_FORCE_LOAD") or os.environ.get("ROCPROF_OUTPUT_PATH"):
            raise RuntimeError("profiler run: rocm-smi skipped")
        p = subprocess.run(["/opt/rocm/bin/rocm-smi", "-d", str(dev), "--showclocks", "--showmaxpower", "--showserial", "--json"],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=20)
        j = json.loads(p.stdout.decode())
        card = j.get("card%d" % dev) or (list(j.values())[0] if j else {})
        keep = {}
        for k, val in card.items():
            kl = k.lower()
            if "sclk" in kl or "mclk" in kl or "fclk" in kl or "power" in kl or "serial" in kl:
                keep[k] = val
        if keep:
            out["rocm_smi"] = keep
    except Exception:
        pass
    return out


# ------------------------------------------------------------------ ranks
def spawn_ranks(n, argv, comm_timeout):
    """`bench.py --gpus N` without a launcher: start the N rank processes (one per GPU) ourselves.  This parent never
    touches a GPU; rank 0's JSON line goes straight to our stdout, every rank's stderr (its device line, RCCL's
    NCCL_DEBUG=WARN output) into a file of its own that is printed when the run fails.  Any rank failing -- or the ranks not
    having formed their communicator within comm_timeout seconds -- ends the run: the children we started are terminated
    and the exit code is non-zero."""
    key = "b%s" % uuid.uuid4().hex[:16]
    logdir = tempfile.mkdtemp(prefix="mcmcx_bench_")
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MCMCX_COMM_KEY=key, MCMCX_BENCH_LOGDIR=logdir)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL needs on this driver
        env.setdefault("NCCL_DEBUG", "WARN")
        lf = open(os.path.join(logdir, "rank%d.stderr" % r), "wb")
        logs.append(lf)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stderr=lf))
    rc = 0
    why = ""
    alive = list(procs)
    t0 = time.time()
    formed = False
    while alive:
        time.sleep(0.05)
        if not formed:
            formed = all(os.path.exists(os.path.join(logdir, "rank%d.formed" % r)) for r in range(n))
            if not formed and time.time() - t0 > comm_timeout and rc == 0:
                rc, why = 3, "the %d ranks had not formed their communicator after %.0f s" % (n, comm_timeout)
                for q in alive:
                    q.terminate()
        for p in list(alive):
            r = p.poll()
            if r is None:
                continue
            alive.remove(p)
            if r != 0 and rc == 0:
                rc, why = r, "rank %d exited with code %d" % (procs.index(p), r)
                for q in alive:                                    # exactly the children we started
                    q.terminate()
    for lf in logs:
        lf.close()
    for r in range(n):
        try:
            txt = open(os.path.join(logdir, "rank%d.stderr" % r), "rb").read().decode(errors="replace")
        except OSError:
            txt = ""
        if rc != 0 or os.environ.get("MCMCX_BENCH_VERBOSE"):
            sys.stderr.write("---- rank %d stderr ----\n%s\n" % (r, txt[-6000:]))
        else:                                                      # success: the device lines only
            sys.stderr.write("".join(l + "\n" for l in txt.splitlines() if l.startswith("bench.py rank")))
    if rc != 0:
        sys.stderr.write("bench.py: %s -- %d ranks could not be formed / run\n" % (why, n))
    import shutil
    shutil.rmtree(logdir, ignore_errors=True)
    return 1 if rc != 0 else 0


def launcher_key():
    """Key of the communicator's bootstrap segment under a launcher (torchrun): the same on every rank of this run, never
    the same for two runs -- the launcher's pid and its start time (field 22 of /proc/<pid>/stat) beside run id and port."""
    ppid = os.getppid()
    try:
        start = open("/proc/%d/stat" % ppid).read().rsplit(")", 1)[1].split()[19]
    except Exception:
        start = "0"
    return "t%s_%s_%d_%s" % (os.environ.get("TORCHELASTIC_RUN_ID", "x"), os.environ.get("MASTER_PORT", "0"), ppid, start)


class Watchdog:
    """Ends THIS process (exit code 3, after a message) if disarm() is not called within `seconds`: around the
    communicator's formation, where a missing peer would otherwise hold ncclCommInitRank forever."""

    def __init__(self, seconds, what):
        self.ev = threading.Event()
        self.t = threading.Thread(target=self._run, args=(seconds, what), daemon=True)
        self.t.start()

    def _run(self, seconds, what):
        if not self.ev.wait(seconds):
            sys.stderr.write("bench.py rank %s: %s did not complete within %.0f s -- giving up (exit 3)\n" % (os.environ.get("RANK", "0"), what, seconds))
            sys.stderr.flush()
            os._exit(3)

    def disarm(self):
        self.ev.set()


# ------------------------------------------------------------------ one configuration
def run_config(wl, steps, warmup, rank, world, dev, comm, chains_per_gpu=0, its_per_step=0, method_opt=None, pooled=False,
               replicas=False, start="default", transport="one GPU", scam_fast=False):
    """Time `steps` bench steps of configuration `wl` (after `warmup` untimed ones) on this rank's GPU; returns the pieces
    of the JSON line on rank 0 (None elsewhere) and the pooled moment vector."""
    from mcmcf90_amd import engine_from_problem
    from mcmcf90_amd.workloads import problem
    n_local = chains_per_gpu or default_chains(wl, world)
    ips = its_per_step or (10 if wl == "c5" else 100)       # c5: one iteration is d = 200 componentwise proposals
    nsimu = (warmup + steps) * ips + 1                       # steps end ON the adaptation ticks (iteration k * ips): one launch + one tick each
    adaptint_ = max(ips, 100)
    it_end = (warmup + steps) * ips
    # a configuration whose timed iterations hold no adaptation (c5: ten iterations per step, adaptint 100) runs on to its next one
    # afterwards, untimed for `value`, so that the line can say what an adaptation costs (N = 1 only)
    # (per-chain rotations at npar = 200, where an adaptation is one SVD per chain, time theirs on a second engine below: the first
    #  adaptation of this line has 100 rows for a 200 x 200 covariance -- rank-deficient, the pinned routine at its 60-sweep cap)
    tick_probe = (world == 1 and it_end // adaptint_ == (warmup * ips) // adaptint_ and not (wl == "c5" and replicas))
    next_tick = (it_end // adaptint_ + 1) * adaptint_
    if tick_probe:
        nsimu = next_tick + 1
    ckw, pkw, per_it = problem(wl, nsimu, adaptint=adaptint_)
    d = pkw["npar"]
    if wl == "c4" and start == "target":
        pkw = dict(pkw, cmat0=np.linalg.inv(np.asarray(pkw["lam"], dtype=float)))
    if wl == "c4" and method_opt == "dram":
        ckw = dict(ckw, method="dram")
    if wl == "c5":
        pooled = not replicas
    elif pooled:
        ckw = dict(ckw, drscale=0.0)                   # c4 keeps method='ram' (pooled RAM); c2 / c3: pooled AM without DR
        if wl == "c4" and method_opt == "dram":
            ckw = dict(ckw, method="dram")
    method = ckw.get("method", "dram")
    eng = engine_from_problem(ckw, pkw, nchains=n_local, chain_id0=rank * n_local, device=dev,
                              pooled=1 if pooled else 0, comm=comm, scam_fast=1 if (scam_fast and wl == "c5") else 0)
    eng.init()

    def one_step(k):
        eng.run((k + 1) * ips)                         # iterations k*ips + 1 .. (k+1)*ips (the very first step starts at iteration 2)
        eng.allreduce_moments(fetch=False)             # local fixed tree -> RCCL all-gather -> tree over ranks; stays in HBM

    def fence():
        eng.sync()                                     # everything this rank queued, the gathers included
        if comm is not None:
            comm.barrier()
            eng.sync()

    for k in range(warmup):
        one_step(k)
    fence()
    eng.kernel_time(reset=True)
    t0 = time.perf_counter()
    for k in range(warmup, warmup + steps):
        one_step(k)
    fence()
    dt = time.perf_counter() - t0
    kms, klaunch, ksteps = eng.kernel_time()
    tot = eng.totals()
    red = np.array([dt, kms]), np.array([float(tot["drtries"]), float(tot["stayed"]), float(tot["downdates"])])
    if comm is not None:
        red = comm.allreduce(red[0], op="max"), comm.allreduce(red[1], op="sum")
    dt, kms = float(red[0][0]), float(red[0][1])                                    # slowest rank
    tries, stayed_all, downs_all = (float(x) for x in red[1])
    pooled_vec = eng.allreduce_moments(fetch=True)                                  # collective: every rank
    kname = eng.last_kernel()
    tick_s = None
    if tick_probe and ckw.get("method", "dram") != "ram":
        eng.run(next_tick - 1); fence()
        tt0 = time.perf_counter()
        eng.run(next_tick); fence()                                                 # one iteration and the adaptation behind it
        tick_s = max(time.perf_counter() - tt0 - dt / (steps * ips), 0.0)
    eng.close()
    tick_note = None
    first_tick_s = second_tick_s = None
    if wl == "c5" and replicas and not scam_fast and world == 1 and steps * ips < adaptint_:
        # per-chain rotations at npar = 200: an adaptation is one SVD per chain.  Timed on a second engine of the same chains whose
        # covariance is full rank at the tick (initcmatn = npar: cmat0 carries weight; adaptint 10) -- what every adaptation after the
        # first is; the configuration's own first one (iteration 100 < npar rows: rank-deficient, the pinned routine runs to its
        # 60-sweep cap) costs about five of these (tools/svd_tick_probe.py, PROBE_INITCMATN=0)
        ckw2, pkw2, _ = problem(wl, 12, adaptint=10)
        ckw2 = dict(ckw2, initcmatn=int(pkw2["npar"]))
        e2 = engine_from_problem(ckw2, pkw2, nchains=n_local, chain_id0=rank * n_local, device=dev, pooled=0, comm=None, scam_fast=0)
        e2.init(); e2.run(9); e2.sync()
        tt0 = time.perf_counter(); e2.run(10); e2.sync()
        tt1 = time.perf_counter(); e2.run(11); e2.sync()
        tt2 = time.perf_counter()
        e2.close()
        tick_s = max((tt1 - tt0) - (tt2 - tt1), 0.0)
        # ... and the adaptations the configuration REALLY starts with: iteration 100 (100 rows for a 200 x 200 covariance:
        # rank 99, the pinned Jacobi runs to its 60-sweep cap on the null space's noise -- so does the oracle) and iteration 200 (rank 199).  A third
        # engine on the configuration's own schedule (adaptint 100, initcmatn 0) gets there with the opt-in fast proposals -- the same accept
        # decisions, states to 1e-13 of the reference-order ones (tests/test_gpu_scam_fast.py), 2.8 s per hundred iterations instead of 160 --
        # and runs the same adaptation kernels.
        ckw3, pkw3, _ = problem(wl, 202, adaptint=adaptint_)
        e3 = engine_from_problem(ckw3, pkw3, nchains=n_local, chain_id0=rank * n_local, device=dev, pooled=0, comm=None, scam_fast=1)
        e3.init()
        early = []
        for tk in (adaptint_, 2 * adaptint_):
            e3.run(tk - 1); e3.sync()
            tt0 = time.perf_counter(); e3.run(tk); e3.sync()
            tt1 = time.perf_counter(); e3.run(tk + 1); e3.sync()
            tt2 = time.perf_counter()
            early.append(max((tt1 - tt0) - (tt2 - tt1), 0.0))
        e3.close()
        first_tick_s, second_tick_s = early
        tick_note = ("`value` is the rate between two adaptations; an adaptation is one 200 x 200 Jacobi SVD per chain.  tick_ms: a full-rank "
                     "covariance (every adaptation from the third on), timed on a second engine of the same %d chains (initcmatn = npar, adaptint 10: "
                     "iteration 10 with its adaptation minus iteration 11).  first_tick_ms / second_tick_ms: the configuration's own adaptations at "
                     "iterations %d and %d (rank-deficient covariances of %d / %d rows: the pinned routine at its sweep cap), timed on a third engine "
                     "that reached them with scam_fast proposals.  sustained_value = the proposals of the configuration's first %d iterations / "
                     "(their time at `value` + those two adaptations + %d full-rank ones)"
                     % (n_local, adaptint_, 2 * adaptint_, adaptint_, 2 * adaptint_, 10 * adaptint_, 8))
    if rank != 0:
        return None, pooled_vec
    its_timed = steps * ips - (1 if warmup == 0 else 0)                         # iteration 1 is the starting point (MCMC_run.F90:35-41)
    its_all = (warmup + steps) * ips - 1
    dr_per_it = tries / its_all                                                 # stage-2 proposals per iteration, all ranks
    proposals = (float(world) * n_local * per_it + dr_per_it) * its_timed
    value = proposals / dt
    per_launch_prop = proposals / world / max(klaunch, 1)                        # proposals one launch of one GPU evaluates
    avg_launch_s = kms / 1e3 / max(klaunch, 1)
    ckey = "%s_%s%s" % (wl, "pooled" if pooled else method, "_target" if start == "target" else "")
    pmc = measured_counters(ckey)
    if wl == "c5" and scam_fast:
        ckey += "_fast"
        pmc = measured_counters(ckey)
    if wl == "c5" and (pooled or scam_fast):    # shared rotation: three d x d products per proposal on the f64 matrix cores
        flop = (2.0 if scam_fast else 6.0) * d * d  # (scam_fast, pooled or per-chain: the target's product only)
        achieved = flop * per_launch_prop / avg_launch_s / 1e12
        roof = {"bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": achieved / FP64_MFMA_PEAK_TF,
                "traffic": pmc["hbm_bytes_per_proposal"] * per_launch_prop if pmc and "hbm_bytes_per_proposal" in pmc else None,
                "kernel": "mcx::" + (kname or "scam_pooled_kernel"), "alg_flop_per_proposal": flop}
    else:
        down_frac = downs_all / (float(world) * n_local * its_all) if method == "ram" else 0.0
        balg = alg_bytes_per_proposal(d, "pooled" if pooled else method, down_frac)
        if method == "scam" and scam_fast:
            balg = 16 * d + 32 + 8 * d                   # scam_fast: one column of the rotation per proposal instead of two passes over it
        achieved = balg * per_launch_prop / avg_launch_s / 1e9
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc["hbm_bytes_per_proposal"] * per_launch_prop if pmc and "hbm_bytes_per_proposal" in pmc else None,
                "kernel": "mcx::" + (kname or "step_kernel"),        # as the engine names the kernel it launched
                "alg_bytes_per_proposal": balg}
        if method == "ram":
            roof["downdate_fraction"] = down_frac
            # what the tile-interleaved layout can reach at best: update and downdate lanes share every 64-byte sector,
            # so an iteration in which a wave holds both kinds moves the factor twice in each direction (2R + 2W)
            tri = d * (d + 1) // 2 * 8
            roof["alg_bytes_2r2w_per_proposal"] = 16 * d + 32 + int(round(tri * (2.0 + 2.0 * (1.0 - (1.0 - down_frac) ** 64))))   # share of waves with a downdate lane
    # `traffic` is not measured in this run (PMC needs passes of its own): it is the stored figure of profiles/traffic.json, used only while the
    # engine sources are the ones it was collected with -- the line says where the number comes from, or that there is none for this build
    roof["traffic_source"] = ("profiles/traffic.json[%s]@%s (%s)" % (ckey, pmc["kernels_sha"], pmc.get("profile"))) if pmc else \
        ("none: profiles/traffic.json holds no entry for %s at engine sha %s" % (ckey, kernels_sha()))
    if pmc:
        if "valu_insts_per_proposal" in pmc:      # second roof: vector-instruction issue, from an SQ counter pass
            ips_ach = pmc["valu_insts_per_proposal"] / 64.0 * per_launch_prop / avg_launch_s / 1e9   # wave-instructions/s
            roof["issue"] = {"achieved": ips_ach, "peak": VALU_ISSUE_PEAK_GIPS, "unit": "G wave-instr/s",
                             "frac": ips_ach / VALU_ISSUE_PEAK_GIPS,
                             "valu_insts_per_proposal": pmc["valu_insts_per_proposal"],
                             "valu_busy": pmc.get("valu_busy")}
            if roof["issue"]["frac"] > roof["frac"]:
                roof["bound_by"] = "valu-issue"
    elif ab_switches():
        roof["traffic_note"] = "PMC figures withheld: A/B switches active (%s)" % ", ".join(ab_switches())
    roof.update(launches=int(klaunch), avg_launch_ms=avg_launch_s * 1e3, kernel_share_of_wall=kms / 1e3 / dt)
    if roof["bound"] == "hbm" and (pooled or d <= 20):
        roof["note"] = ("the chip's HBM roof is quoted for uniformity; this configuration is bound by the per-chain random "
                        "numbers (Philox + polar + pinned log/sqrt on the VALU), see DESIGN.md section 5")
    if method == "ram":
        stay = stayed_all / (float(world) * n_local * its_all)
        roof["note"] = ("accepted fraction %.2f (start: %s); RAM updates the factor after alpha >= alphatarget (one "
                        "read+write sweep) and downdates it otherwise (two sweeps): DESIGN.md section 10 item 6" % (1.0 - stay, start))
    mode = method + (" pooled (one shared factor)" if pooled else ", per-chain factor") + \
        (", scam_fast = 1 (opt-in: proposals as theta + delta U(:,j), not the reference's operation order)" if (scam_fast and wl == "c5") else "")
    # what the timed iterations hold besides proposals: the adaptation ticks (every adaptint iterations), the starting regime
    adaptint = int(ckw.get("adaptint", 100))
    nticks = sum(1 for it in range(warmup * ips + 1, (warmup + steps) * ips + 1) if it % adaptint == 0) if method != "ram" else 0
    regime = ""
    if method != "ram":
        regime = "; adaptint = %d: %d adaptation tick(s) inside the %d timed iterations" % (adaptint, nticks, steps * ips)
        if nticks == 0:
            regime += " (the rate between two adaptations)"
            if wl == "c5" and replicas:
                regime += "; an adaptation here is one SVD per chain (`adaptation.tick_ms`: full-rank covariances)"
    elif start == "target":
        regime = "; cmat0 = the target's covariance: RAM at its target acceptance rate from the start (about half of the lanes downdate)"
    adaptation = None
    if tick_s is not None:
        t_it = dt / (steps * ips)
        sustained = float(world) * n_local * per_it * adaptint / (adaptint * t_it + tick_s)
        extra_ticks = {}
        if first_tick_s is not None:                     # the configuration's own schedule over its first ten adaptation periods
            sustained = float(world) * n_local * per_it * 10 * adaptint / (10 * adaptint * t_it + first_tick_s + second_tick_s + 8 * tick_s)
            extra_ticks = {"first_tick_ms": first_tick_s * 1e3, "second_tick_ms": second_tick_s * 1e3}
        adaptation = {"adaptint": adaptint, "tick_ms": tick_s * 1e3, **extra_ticks,
                      "sustained_value": sustained,
                      "note": tick_note or ("`value` is the rate between two adaptations; sustained_value = proposals of adaptint iterations / (their time + one "
                                            "adaptation), the adaptation timed once after the timed region (iteration %d)" % next_tick)}
    cnt = float(pooled_vec[0])
    mean = pooled_vec[1:1 + d] / cnt
    res = {
        "metric": "MH proposals/sec (whole node), d=50 Gaussian target" if wl == "c4" else "MH proposals/sec (whole node), " + WORKLOADS[wl],
        "value": value, "ms_per_step": dt / steps * 1e3,
        "config": {"workload": "%s, method=%s, %d chains/GPU, pooled moments of all %d chains combined every %d iterations (%s)%s"
                               % (WORKLOADS[wl], mode, n_local, world * n_local, ips, transport, regime),
                   "chains_per_gpu": n_local, "total_chains": world * n_local, "its_per_step": ips, "npar": d, "method": method,
                   "proposals_per_iteration": per_it + dr_per_it / (float(world) * n_local),
                   "parallelism": "chains sharded over %d GPU(s), one process each" % world},
        "roofline": roof,
        "adaptation": adaptation,
        "pooled_check": {"chains": cnt, "max_abs_mean": float(np.max(np.abs(mean)))},
        "_cpu": (wl, ckw, pkw, per_it, "d=%d %s" % (d, method)),
    }
    return res, pooled_vec


# the other BASELINE configurations, run briefly after the headline in the same process (N = 1): (key, run_config arguments)
OTHER_CONFIGS = [
    # (printed in this order, and the driver keeps the LAST 2000 characters of stdout: the BASELINE configurations come last)
    # the few-chains regime (one tile of 64 chains: an iteration is latency, not throughput): us per iteration incl. the adaptation ticks
    ("c2_64_chains", dict(wl="c2", steps=5, warmup=1, chains_per_gpu=64)),
    ("c3_64_chains", dict(wl="c3", steps=5, warmup=1, chains_per_gpu=64)),
    ("c4_64_chains", dict(wl="c4", steps=5, warmup=1, chains_per_gpu=64)),       # method = 'ram' with one tile: group_ram_kernel (factor in registers)
    ("c5_pooled_scam_fast", dict(wl="c5", steps=2, warmup=1, scam_fast=True)),        # opt-in variants, labelled as such in `workload`
    ("c5_replicas_scam_fast", dict(wl="c5", steps=2, warmup=1, replicas=True, scam_fast=True)),
    # config 1's model at scale: one response column (the lane-group kernels) and two (nycol = 2: step_kernel_cols) -- VERDICT round 5, item 6
    ("c1", dict(wl="c1", steps=3, warmup=1)),
    ("c1x", dict(wl="c1x", steps=3, warmup=1)),
    # MCMC_run_scam.F90:106-115 as written: per-chain rotations, two dgemv per componentwise proposal (no scam_fast); one iteration per step
    ("c5_replicas", dict(wl="c5", steps=2, warmup=1, replicas=True, its_per_step=1)),
    ("c5_pooled", dict(wl="c5", steps=2, warmup=1)),
    ("c4_pooled", dict(wl="c4", steps=6, warmup=1, pooled=True)),
    ("c4_target", dict(wl="c4", steps=3, warmup=1, start="target", chains_per_gpu=131072)),    # one eighth of the chains: 0.14 s per step
    ("c3", dict(wl="c3", steps=3, warmup=1)),
    ("c2", dict(wl="c2", steps=10, warmup=2)),                      # 1.7 ms per step: ten of them
]
# the configurations BASELINE.json names (and c4's two other regimes): mirrored as [value, roofline fraction] into `roofline.others`
HEADLINE_OTHERS = ("c2", "c3", "c4_target", "c4_pooled", "c5_pooled", "c5_replicas", "c1x")
# N > 1: the pooled form of the headline configuration on the same communicator -- its RAM tick (the rank-one statistics of all chains
# of all ranks gathered and folded into the one shared factor every adaptint iterations) is the collective ON the critical path
OTHER_CONFIGS_MULTI = [
    ("c4_pooled", dict(wl="c4", steps=6, warmup=1, pooled=True)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS), help="BASELINE.json configuration (default: the headline one)")
    ap.add_argument("--chains-per-gpu", type=int, default=0, help="default: c4 / c5: the configuration's chains divided by --gpus (c4: 1048576 / N, strong scaling); c2 / c3: the configuration's count per GPU")
    ap.add_argument("--its-per-step", type=int, default=0, help="MH iterations per bench step (default 100; c5: 10)")
    ap.add_argument("--method", default=None, choices=["ram", "dram"], help="c4 only: per-chain RAM (default) or AM")
    ap.add_argument("--pooled", action="store_true", help="one shared factor from the all-reduced pooled covariance (c5: the default)")
    ap.add_argument("--replicas", action="store_true", help="c5: per-chain rotations (the reference's semantics) instead of the pooled one")
    ap.add_argument("--scam-fast", action="store_true", help="c5: the opt-in fast componentwise proposal (mcmcx_config::scam_fast); never the default")
    ap.add_argument("--start", default="default", choices=["default", "target"],
                    help="c4: 'target' starts from cmat0 = Sigma, i.e. at RAM's target acceptance rate, where most iterations are "
                         "Cholesky downdates (default: cmat0 = 0.01 I, 86 %% accepted, RAM adapts by updates)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of the other BASELINE configurations after the headline")
    ap.add_argument("--comm-timeout", type=float, default=float(os.environ.get("MCMCX_COMM_TIMEOUT", "240")),
                    help="seconds the ranks may take to form their communicator before the run is given up")
    ap.add_argument("--allow-host-transport", action="store_true",
                    help="N > 1: if RCCL refuses to form although every rank has its own GPU, carry the (latency-sized) exchange through the host "
                         "segment and say so in the line (`transport`: host, `rccl_ranks` 0).  Off by default: an RCCL failure then ends the run "
                         "with a non-zero exit code and RCCL's own NCCL_DEBUG=WARN output, so that a scaling record cannot hold a non-RCCL number unasked")
    ap.add_argument("--verbose", action="store_true", help="`other_configs` entries in their long form (prose workload strings, every field) instead of the compact one")
    ap.add_argument("--one-gpu-dryrun", action="store_true",
                    help="debug: all ranks share GPU 0 and exchange through the host transport (checks the N>1 path on a 1-GPU box)")
    ap.add_argument("--dump-moments", default=None, help="debug: rank 0 writes the final pooled moment vector (float64) to this file")
    ap.add_argument("--plan", action="store_true", help="print how --gpus N would shard the configuration (one JSON line: chains and first chain id of "
                                                        "every rank, scaling mode) and exit; touches no GPU and starts no rank")
    a = ap.parse_args()
    if a.plan:
        print(json.dumps(shard_plan(a.workload, a.gpus, a.chains_per_gpu)), flush=True)
        return

    if "WORLD_SIZE" not in os.environ:
        if a.gpus < 1:
            raise SystemExit("--gpus must be >= 1")
        if a.gpus > 1:
            sys.exit(spawn_ranks(a.gpus, sys.argv[1:], a.comm_timeout))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: one rank per GPU" % (a.gpus, world))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this driver
    if world > 1:
        os.environ.setdefault("NCCL_DEBUG", "WARN")                # RCCL's own diagnostics on stderr, should the formation fail
    from mcmcf90_amd import Comm, _lib
    L = _lib.load()
    ndev = L.mcmcx_device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    dev = 0 if a.one_gpu_dryrun else local_rank
    if dev >= ndev:
        if ndev == 1 and world > 1 and not a.one_gpu_dryrun and any(os.environ.get(v) for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")):
            dev = 0                                   # the launcher gave every rank its own single visible GPU
        else:
            raise SystemExit("rank %d: local rank %d but %d HIP device(s) visible -- %d ranks need %d GPUs (or --one-gpu-dryrun)" % (rank, local_rank, ndev, world, world))
    import ctypes
    info = ctypes.create_string_buffer(256)
    L.mcmcx_device_info(dev, info, 256)
    comm = None
    rccl_ranks = 1
    if world > 1:
        key = os.environ.get("MCMCX_COMM_KEY") or launcher_key()
        wd = Watchdog(a.comm_timeout, "forming the %d-rank communicator (%s)" % (world, "host transport" if a.one_gpu_dryrun else "ncclCommInitRank"))
        rccl_failed = None
        simulate = bool(os.environ.get("MCMCX_BENCH_SIMULATE_RCCL_FAILURE"))      # (tests: what an RCCL failure does, on a one-GPU box)
        try:
            if simulate:
                raise RuntimeError("simulated: ncclCommInitRank failed")
            comm = Comm(key, rank, world, dev, backend="host" if a.one_gpu_dryrun else "rccl")    # raises unless all ranks arrive
            comm.barrier()
        except Exception as ex:
            # RCCL would not form (it has never met N > 1 GPUs in this project's own runs).  The path's messages are latency-sized (timing scalars,
            # one moment vector per tick), so the same exchange staged through the shared-memory segment gives numbers that stand -- but only
            # with a GPU per rank (ranks sharing a device must stay an error: that is what RCCL refuses first, and a scaling point measured that
            # way would be a lie) and only when asked for:
            # Opt-in (--allow-host-transport): by default an RCCL failure is fatal -- rank stderr carries NCCL_DEBUG=WARN's lines, the exit code
            # is non-zero, no JSON line is printed.  With the flag EVERY rank whose formation raised falls back, whatever its message said (a rank
            # that only saw a peer's `failed` word or timed out in the bootstrap gets no "nccl" in its text; deciding per message made the ranks
            # disagree -- ADVICE round 5): a peer that is really gone fails the host communicator's formation the same way.
            if (not simulate and (a.one_gpu_dryrun or ndev < world)) or not a.allow_host_transport:
                sys.stderr.write("bench.py rank %d: forming the %d-rank RCCL communicator failed: %r\n"
                                 "bench.py rank %d: no number is produced (--allow-host-transport would carry the exchange through the host segment)\n"
                                 % (rank, world, ex, rank))
                sys.stderr.flush()
                raise SystemExit(4)
            rccl_failed = repr(ex)[:300]
            sys.stderr.write("bench.py rank %d: RCCL communicator failed (%s); falling back to the host transport\n" % (rank, rccl_failed))
            sys.stderr.flush()
            comm = Comm(key + "h", rank, world, dev, backend="host")
            comm.barrier()
        wd.disarm()
        rccl_ranks = int(L.mcmcx_comm_size(comm.h))
        if rccl_ranks != world or int(L.mcmcx_comm_rank(comm.h)) != rank:
            raise SystemExit("rank %d: the communicator reports rank %d of %d, expected %d of %d" % (rank, L.mcmcx_comm_rank(comm.h), rccl_ranks, rank, world))
        ld = os.environ.get("MCMCX_BENCH_LOGDIR")
        if ld:
            open(os.path.join(ld, "rank%d.formed" % rank), "w").close()
    sys.stderr.write("bench.py rank %d/%d: HIP device %d of %d visible [%s], communicator %s\n"
                     % (rank, world, dev, ndev, info.value.decode(errors="replace"),
                        "none (one GPU)" if comm is None else "%s rank %d of %d" % ("host-staged" if a.one_gpu_dryrun else "RCCL", rank, rccl_ranks)))
    sys.stderr.flush()
    rccl_failed = rccl_failed if world > 1 else None
    transport = "one GPU" if world == 1 else (("host transport (RCCL did not form: %s)" % rccl_failed) if rccl_failed else
                                               "host transport, ranks share GPU 0" if a.one_gpu_dryrun else "RCCL all-gather + fixed tree")
    res, pooled_vec = run_config(a.workload, a.steps, a.warmup, rank, world, dev, comm, chains_per_gpu=a.chains_per_gpu,
                                 its_per_step=a.its_per_step, method_opt=a.method, pooled=a.pooled, replicas=a.replicas,
                                 start=a.start, transport=transport, scam_fast=a.scam_fast)
    plain = a.workload == "c4" and not (a.pooled or a.method or a.start != "default" or a.its_per_step or a.replicas or a.scam_fast)
    line = None
    if rank == 0:
        cpu_args = res.pop("_cpu")
        line = {
            "metric": res["metric"], "value": res["value"], "unit": "proposals/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True,
            "scaling": "strong" if (a.workload in STRONG and not a.chains_per_gpu) else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": res["config"], "roofline": res["roofline"], "pooled_check": res["pooled_check"],
            **({"adaptation": res["adaptation"]} if res.get("adaptation") else {}),
            # what carried the N > 1 exchange: "rccl", "host" (opt-in fallback or --one-gpu-dryrun; rccl_ranks 0), "none" (one GPU)
            "transport": "none" if world == 1 else ("host" if (a.one_gpu_dryrun or rccl_failed) else "rccl"),
            "rccl_ranks": rccl_ranks if (world > 1 and not a.one_gpu_dryrun and not rccl_failed) else (1 if world == 1 else 0),
            **({"rccl_error": rccl_failed} if rccl_failed else {}),
            "engine_sha": kernels_sha(),
            "device": device_ident(L, dev),
        }
    others, details = {}, {}
    if not a.no_other_configs and plain and (world > 1 or not a.chains_per_gpu):
        # N = 1: the other BASELINE configurations, briefly, in this same process (builder-independent numbers for all of them);
        # N > 1: the pooled form of the headline on the same communicator (every rank runs it: its ticks are collective, so an
        # error there is not swallowed -- a rank that raised alone would leave its peers in the next gather)
        for key, kw in (OTHER_CONFIGS if world == 1 else OTHER_CONFIGS_MULTI):
            try:
                t0 = time.perf_counter()
                r, _ = run_config(rank=rank, world=world, dev=dev, comm=comm, transport=transport,
                                  **dict(kw, **({"chains_per_gpu": a.chains_per_gpu} if (world > 1 and a.chains_per_gpu) else {})))
                if rank == 0:
                    rf = r["roofline"]
                    details[key] = {"value": r["value"], "unit": "proposals/s", "ms_per_step": r["ms_per_step"], "steps": kw["steps"],
                                    "bound": rf["bound"], "roofline_frac": rf["frac"], "achieved": rf["achieved"], "roofline_unit": rf["unit"],
                                    "issue_frac": rf.get("issue", {}).get("frac"), "kernel": rf["kernel"],
                                    "avg_launch_ms": rf["avg_launch_ms"], "kernel_share_of_wall": rf["kernel_share_of_wall"],
                                    "alg_per_proposal": rf.get("alg_bytes_per_proposal", rf.get("alg_flop_per_proposal")),
                                    "traffic_source": rf.get("traffic_source"),
                                    "workload": r["config"]["workload"], "proposals_per_iteration": r["config"]["proposals_per_iteration"],
                                    "chains_per_gpu": r["config"]["chains_per_gpu"], "n_gpus": world,
                                    "us_per_iteration": r["ms_per_step"] * 1e3 / r["config"]["its_per_step"],
                                    "wall_s_incl_init": time.perf_counter() - t0}
                    # the compact entry (<= 160 characters): value, fraction of the roof named in `bound`, the vector-issue fraction where an SQ pass
                    # exists for this build, the kernel as the engine names it, ms per step, the kernel's share of the wall; few chains: us per iteration
                    c = {"value": float("%.4g" % r["value"]), "frac": round(rf["frac"], 4), "bound": rf["bound"]}
                    if rf.get("issue"):
                        c["issue"] = round(rf["issue"]["frac"], 3)
                    c.update(kernel=rf["kernel"].replace("mcx::", ""), ms=float("%.4g" % r["ms_per_step"]), share=round(rf["kernel_share_of_wall"], 3))
                    if r["config"]["chains_per_gpu"] <= 64:
                        c["us_it"] = float("%.3g" % details[key]["us_per_iteration"])
                    if world > 1:
                        details[key]["rccl_ranks"] = line["rccl_ranks"]
                        details[key]["pooled_check"] = r["pooled_check"]
                        c["rccl_ranks"] = line["rccl_ranks"]
                    if r.get("adaptation"):
                        details[key].update(tick_ms=r["adaptation"]["tick_ms"], sustained_value=r["adaptation"]["sustained_value"])
                        c["tick_ms"] = float("%.4g" % r["adaptation"]["tick_ms"])
                        for k2 in ("first_tick_ms", "second_tick_ms"):
                            if k2 in r["adaptation"]:
                                details[key][k2] = r["adaptation"][k2]
                                c[k2] = float("%.4g" % r["adaptation"][k2])
                    others[key] = c
            except Exception as ex:
                if world > 1:
                    raise
                others[key] = {"error": str(ex)[:120]}
                details[key] = {"error": str(ex)[:300]}
    if rank == 0:
        if others:
            # the driver stores `roofline` whole: every BASELINE configuration's [proposals/s, fraction of its roof] rides there too
            line["roofline"]["others"] = {k: [others[k]["value"], others[k]["frac"]] for k in HEADLINE_OTHERS if k in others and "value" in others[k]}
        if not a.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(*cpu_args)
        if others:
            line["other_configs"] = details if a.verbose else others       # LAST key of the line: the tail of stdout holds every configuration
            try:                                                          # the long form (prose workload strings, every field) beside the line
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                with open(os.path.join(ROOT, "gpurun_out", "bench_other_configs.json"), "w") as fh:
                    json.dump(details, fh, indent=1)
            except OSError:
                pass
        if a.dump_moments:
            np.asarray(pooled_vec, dtype=np.float64).tofile(a.dump_moments)
        print(json.dumps(line), flush=True)
    if comm is not None:
        comm.barrier()
        comm.close()


if __name__ == "__main__":
    main()
