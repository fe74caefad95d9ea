"""N > 1 through the engine, on the one GPU of the test box: bench.py starts two rank
processes that share GPU 0 and exchange through libmcmcx.so's host transport (RCCL refuses two ranks on one device);
everything else -- sharding by chain_id0, the local fixed tree, the gather, the tree over ranks, the pooled-mode
adaptation from the moments of ALL ranks -- is the code an 8-GPU run executes.  Fixed-tree claim of DESIGN.md
section 7: the pooled moments of 2 ranks x n chains are bit-identical to 1 rank x 2n chains."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, tmp_path, name):
    dump = str(tmp_path / (name + ".f64"))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MCMCX_COMM_KEY")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--dump-moments", dump] + args,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()
    return json.loads(lines[0]), np.fromfile(dump, dtype=np.float64)


@pytest.mark.parametrize("extra", [[], ["--pooled"], ["--pooled", "--method", "dram"], ["--workload", "c2"],
                                   ["--workload", "c5", "--chains-per-gpu", "128", "--its-per-step", "2"]],
                         ids=["c4_ram", "c4_pooled_ram", "c4_pooled_am", "c2_am", "c5_pooled_scam"])
def test_two_ranks_equal_one_rank_with_twice_the_chains(tmp_path, extra):
    n = 2048
    common = ["--steps", "2", "--warmup", "1"] + extra
    if "--chains-per-gpu" in extra:
        n = int(extra[extra.index("--chains-per-gpu") + 1])
        common = [a for i, a in enumerate(common) if a != "--chains-per-gpu" and (i == 0 or common[i - 1] != "--chains-per-gpu")]
    two, m2 = _bench(common + ["--gpus", "2", "--one-gpu-dryrun", "--chains-per-gpu", str(n)], tmp_path, "two")
    one, m1 = _bench(common + ["--gpus", "1", "--chains-per-gpu", str(2 * n)], tmp_path, "one")
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["pooled_check"]["chains"] == 2 * n == one["pooled_check"]["chains"]
    assert two["config"]["chains_per_gpu"] == n
    assert m2.shape == m1.shape and m2[0] == 2 * n
    assert np.array_equal(m2.view(np.uint64), m1.view(np.uint64)), "pooled moments of 2 ranks differ from 1 rank with twice the chains"
    # whole-job rate: both ranks' proposals over the slowest rank's time
    assert two["value"] > 0 and two["roofline"]["launches"] == one["roofline"]["launches"]


def test_two_ranks_cannot_form_on_one_gpu_without_the_dry_run(tmp_path):
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    n = ctypes.c_int(0)
    hip.hipGetDeviceCount(ctypes.byref(n))
    if n.value >= 2:
        pytest.skip("two GPUs are present")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MCMCX_COMM_KEY")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--chains-per-gpu", "64",
                        "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
    assert p.returncode != 0
    assert not [l for l in p.stdout.decode().splitlines() if l.startswith("{")]


def test_rccl_communicator_of_one_rank(tmp_path):
    """The RCCL transport itself (ncclGetUniqueId through the shm bootstrap, ncclCommInitRank, ncclAllReduce for the
    host scalars, the engine's all-gather + tree) with the one rank a one-GPU box can form: results equal the
    communicator-less engine bit for bit."""
    import uuid
    from mcmcf90_amd import Comm, engine_from_problem
    from mcmcf90_amd.workloads import problem
    comm = Comm("t" + uuid.uuid4().hex[:12], 0, 1, 0, backend="rccl")
    assert comm.allreduce(np.array([3.0, -1.5]), op="sum").tolist() == [3.0, -1.5]
    assert comm.allreduce(np.array([3.0, -1.5]), op="max").tolist() == [3.0, -1.5]
    comm.barrier()
    ckw, pkw, _ = problem("c2", 301)
    ckw = dict(ckw, drscale=0.0)
    res = []
    for c in (comm, None):
        e = engine_from_problem(ckw, pkw, nchains=256, pooled=1, comm=c)
        e.init(); e.run()
        res.append((e.allreduce_moments(), e.pooled_moments(), e.theta()))
        e.close()
    comm.close()
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    assert np.array_equal(res[0][0].view(np.uint64), res[0][1].view(np.uint64))      # one rank: all-reduced == local


def test_one_process_node_api_with_one_gpu():
    """mcmcx_comm_create_all / mcmcx_run_all / mcmcx_allreduce_moments_all (what the Fortran shim's `ngpus` drives) with
    the one device of the box: ncclCommInitAll(1), the threaded run and the grouped gather are exercised."""
    import ctypes as C
    from mcmcf90_amd import _lib, engine_from_problem
    from mcmcf90_amd.workloads import problem
    L = _lib.load()
    comms = (C.c_void_p * 1)()
    assert L.mcmcx_comm_create_all(1, None, comms) == 0, L.mcmcx_last_error()
    ckw, pkw, _ = problem("c2", 201)

    class _C:                                    # duck-typed Comm for Engine.set_comm
        h = C.c_void_p(comms[0])
    e = engine_from_problem(ckw, pkw, nchains=128, pooled=1, comm=_C())
    e.init()
    hs = (C.c_void_p * 1)(e.h)
    assert L.mcmcx_run_all(hs, 1, 201) == 0, L.mcmcx_last_error()
    out = np.zeros(L.mcmcx_pooled_moments_len(e.h))
    assert L.mcmcx_allreduce_moments_all(hs, 1, out.ctypes.data_as(C.POINTER(C.c_double))) == 0, L.mcmcx_last_error()
    assert np.array_equal(out.view(np.uint64), e.pooled_moments().view(np.uint64)) and out[0] == 128
    assert e.simuind == 201
    e.close()
    L.mcmcx_comm_destroy(comms[0])


def test_two_ranks_under_torchrun(tmp_path):
    """The driver's launch line (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`): ranks from
    RANK / LOCAL_RANK / WORLD_SIZE, communicator key from the launcher's port and pid.  Two ranks on the one GPU
    (dry run), equal to the self-launched form."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MCMCX_COMM_KEY")}
    dump = str(tmp_path / "tr.f64")
    args = ["--gpus", "2", "--one-gpu-dryrun", "--steps", "2", "--warmup", "1", "--chains-per-gpu", "1024", "--no-cpu-baseline"]
    port = 29600 + os.getpid() % 300
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py")] + args + ["--dump-moments", dump],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["pooled_check"]["chains"] == 2048
    own, m_own = _bench(args, tmp_path, "own")
    assert np.array_equal(np.fromfile(dump, dtype=np.float64).view(np.uint64), m_own.view(np.uint64))


def test_rccl_failure_is_fatal_unless_the_host_transport_is_allowed(tmp_path, monkeypatch):
    """RCCL has never met N > 1 GPUs in this project's own runs (no node was ever available).  Should its communicator fail to form on the
    driver's node, bench.py must NOT produce a quiet number: by default the run ends non-zero, prints no JSON line
    and says why on stderr.  With --allow-host-transport it carries the path's latency-sized messages through the shared-memory segment
    and SAYS so in its line (`transport` host, `rccl_error`, `rccl_ranks` 0, the transport named in the workload).  Simulated here
    (MCMCX_BENCH_SIMULATE_RCCL_FAILURE): two ranks, the same moments as the plain dry run."""
    args = ["--gpus", "2", "--one-gpu-dryrun", "--steps", "2", "--warmup", "1", "--chains-per-gpu", "1024", "--no-other-configs"]
    plain, m_plain = _bench(args, tmp_path, "plain")
    assert plain["transport"] == "host" and plain["rccl_ranks"] == 0            # a dry run never claims RCCL either
    monkeypatch.setenv("MCMCX_BENCH_SIMULATE_RCCL_FAILURE", "1")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MCMCX_COMM_KEY")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600, env=env)
    assert p.returncode != 0, "an RCCL failure without --allow-host-transport must end the run non-zero"
    assert not [l for l in p.stdout.decode().splitlines() if l.startswith("{")], "... and print no JSON line"
    assert "--allow-host-transport" in p.stderr.decode() and "simulated" in p.stderr.decode()
    fb, m_fb = _bench(args + ["--allow-host-transport"], tmp_path, "fallback")
    assert "rccl_error" in fb and "simulated" in fb["rccl_error"] and fb["rccl_ranks"] == 0 and "rccl_error" not in plain
    assert fb["transport"] == "host"
    assert "RCCL did not form" in fb["config"]["workload"] or "RCCL did not form" in json.dumps(fb["config"])
    assert fb["n_gpus"] == 2 and fb["pooled_check"]["chains"] == 2048
    assert np.array_equal(m_fb.view(np.uint64), m_plain.view(np.uint64))


def test_the_whole_gpus_n_json_path_with_as_many_ranks_as_the_box_allows(tmp_path):
    """The complete `bench.py --gpus N` line -- headline configuration AND the N > 1 `other_configs` run (c4 pooled on the same
    communicator: the collective on the critical path) -- with N = 4 rank processes x 64 chains on the one GPU over the host transport
    (the GPU box allows six processes on its card: pytest + 4 ranks + the launcher's margin; N = 8 is covered inside one process by
    test_eight_ranks_in_one_process... and on the host transport without a GPU by tests/test_comm_cpu.py).  The line must say what it is:
    transport host, rccl_ranks 0, total_chains N x 64, the value of N x 64 chains' proposals -- and the headline's pooled moments must be
    those of ONE rank with N x 64 chains bit for bit."""
    world, n = 4, 64
    common = ["--steps", "2", "--warmup", "1"]
    many, mN = _bench(common + ["--gpus", str(world), "--one-gpu-dryrun", "--chains-per-gpu", str(n)], tmp_path, "many")
    one, m1 = _bench(common + ["--gpus", "1", "--chains-per-gpu", str(world * n), "--no-other-configs"], tmp_path, "one")
    assert many["n_gpus"] == world and many["transport"] == "host" and many["rccl_ranks"] == 0
    assert many["config"]["total_chains"] == world * n and many["config"]["chains_per_gpu"] == n
    assert many["pooled_check"]["chains"] == world * n
    its = 2 * many["config"]["its_per_step"]                       # iteration 1 is the starting point only when warmup = 0
    assert abs(many["value"] * many["ms_per_step"] * 2e-3 - world * n * its) < 1e-6 * world * n * its, "value x time != the proposals of N x 64 chains"
    assert np.array_equal(mN.view(np.uint64), m1.view(np.uint64))
    oc = many["other_configs"]["c4_pooled"]
    assert oc["rccl_ranks"] == 0 and oc["value"] > 0 and "pooled" in oc["kernel"], oc
    assert len(json.dumps(oc)) <= 160, "compact entries: the driver keeps the last 2000 characters of stdout"


@pytest.mark.parametrize("nranks", [4])   # the GPU box allows 6 processes on its card: pytest + 4 ranks
def test_tree_over_ranks_equals_tree_over_tiles(tmp_path, nranks):
    """The claim behind all-gather + tree: for power-of-two shards the pairwise tree over ranks
    continues the pairwise tree over tiles, so N ranks x n chains give the bits of 1 rank x N n chains.  Two ranks
    cannot tell a tree from a ring (a + b = b + a); four and eight can."""
    n = 512
    common = ["--steps", "2", "--warmup", "1", "--pooled", "--method", "dram"]     # pooled AM: the moments feed back into every proposal
    many, mN = _bench(common + ["--gpus", str(nranks), "--one-gpu-dryrun", "--chains-per-gpu", str(n)], tmp_path, "many")
    one, m1 = _bench(common + ["--gpus", "1", "--chains-per-gpu", str(nranks * n)], tmp_path, "one")
    assert many["n_gpus"] == nranks and many["pooled_check"]["chains"] == nranks * n
    assert np.array_equal(mN.view(np.uint64), m1.view(np.uint64))


@pytest.mark.parametrize("mode", ["pooled_am", "pooled_ram", "replicas_moments"])
def test_eight_ranks_in_one_process_equal_one_rank_with_eight_times_the_chains(mode):
    """N = 8 -- the node the scaling bench runs on -- through the engine on the one GPU of the box: eight engines of n chains (chain_id0 = r n),
    one host thread each, meet in eight communicators of the host transport formed INSIDE this process (the GPU box allows six processes
    on its card; threads are free), i.e. every tick runs the local tree, the eight-slot gather and moments_tree_kernel with ranks as
    tiles.  The pooled factor, every chain's state and the all-reduced moment vector must be those of ONE engine of 8 n chains bit for
    bit: for the moments of the current states (pooled AM; replicas mode: the posterior-moment output) AND for the pooled-RAM statistic
    (the third kind of exchanged vector).  Eight can tell a pairwise tree from a ring or a linear sum; two cannot."""
    import threading
    import uuid
    from mcmcf90_amd import Comm, engine_from_problem
    from mcmcf90_amd.workloads import problem
    world, n = 8, 128                                       # two tiles per rank
    if mode == "pooled_ram":
        ckw, pkw, _ = problem("c4", 131, adaptint=20)
        ekw = dict(pooled=1)
    elif mode == "pooled_am":
        ckw, pkw, _ = problem("c4", 231, adaptint=50)
        ckw = dict(ckw, method="dram")
        ekw = dict(pooled=1)
    else:
        ckw, pkw, _ = problem("c2", 231)
        ekw = dict()
    one = engine_from_problem(ckw, pkw, nchains=world * n, record_accept=1, **ekw)
    one.init(); one.run()
    ref = dict(theta=one.theta(), mom=one.allreduce_moments(), pooled=one.pooled()[3] if ekw else None, masks=one.accept_masks())
    one.close()
    key = "t8" + uuid.uuid4().hex[:12]
    res, errs = [None] * world, []

    def rank_main(r):
        try:
            c = Comm(key, r, world, 0, backend="host")
            e = engine_from_problem(ckw, pkw, nchains=n, chain_id0=r * n, record_accept=1, comm=c, **ekw)
            e.init(); e.run()
            res[r] = dict(theta=e.theta(), mom=e.allreduce_moments(), pooled=e.pooled()[3] if ekw else None, masks=e.accept_masks())
            c.barrier()
            e.close(); c.close()
        except Exception as ex:                              # a rank that fails must not leave the others in a gather for ever
            errs.append((r, repr(ex)))
            raise

    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs and all(x is not None for x in res), errs
    bits = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
    assert np.array_equal(bits(np.vstack([x["theta"] for x in res])), bits(ref["theta"]))
    assert np.array_equal(np.hstack([x["masks"] for x in res]), ref["masks"])
    for x in res:
        assert np.array_equal(bits(x["mom"]), bits(ref["mom"])) and x["mom"][0] == world * n
        if ekw:
            assert np.array_equal(bits(x["pooled"]), bits(ref["pooled"]))


def test_rocm_rccl_without_torch(tmp_path):
    """bench.py's ranks never import torch, so their librccl / libamdhip64 are /opt/rocm's (pytest's own process has
    torch's copies loaded first).  The same transport calls in a torch-free child: unique id through the shm bootstrap,
    ncclCommInitRank, ncclAllReduce, the engine's all-gather + tree."""
    code = r"""
import sys, uuid, numpy as np
sys.path.insert(0, %r)
assert "torch" not in sys.modules
from mcmcf90_amd import Comm, engine_from_problem
from mcmcf90_amd.workloads import problem
c = Comm("t" + uuid.uuid4().hex[:12], 0, 1, 0, backend="rccl")
assert c.allreduce(np.array([2.5, -1.0]), op="sum").tolist() == [2.5, -1.0]
c.barrier()
ckw, pkw, _ = problem("c2", 201)
e = engine_from_problem(dict(ckw, drscale=0.0), pkw, nchains=128, pooled=1, comm=c)
e.init(); e.run()
a, b = e.allreduce_moments(), e.pooled_moments()
assert np.array_equal(a.view(np.uint64), b.view(np.uint64)) and a[0] == 128
e.close(); c.close()
assert "torch" not in sys.modules
maps = open("/proc/self/maps").read()
print("RCCL_FROM", [l.split()[-1] for l in maps.splitlines() if "librccl" in l][0])
""" % ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MCMCX_COMM_KEY")}
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    assert "RCCL_FROM /opt/rocm" in p.stdout.decode(), p.stdout.decode()


def test_a_signal_caught_by_one_rank_stops_all_ranks_at_the_same_tick(tmp_path):
    """in pooled mode the ranks meet in every tick's gather, so a rank that acted on a signal by itself
    would leave its peers waiting.  The stop decision is collective: the rank that caught SIGUSR1 only raises its flag in
    the exchanged vector, and BOTH ranks leave mcmcx_run with MCMCX_INTERRUPTED at the same adaptation tick -- same
    simuind, same pooled factor (the reference saves the chain "upto simuind" and stops, MCMC_signal_handler.F90:95-107)."""
    import signal
    import time
    import uuid
    key = "sig%s" % uuid.uuid4().hex[:12]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MCMCX_COMM_KEY")}
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multirank_worker.py"), key, str(r), "2", str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(2)]
    try:
        t0 = time.time()
        while not all((tmp_path / ("rank%d.ready" % r)).exists() for r in range(2)):
            assert time.time() - t0 < 240 and all(p.poll() is None for p in procs), [p.stdout.read().decode()[-2000:] for p in procs if p.poll() is not None]
            time.sleep(0.05)
        time.sleep(0.5)
        procs[1].send_signal(signal.SIGUSR1)                 # one rank only
        outs = [p.communicate(timeout=240)[0].decode(errors="replace") for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), outs
    res = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    assert res[0]["rc"] == 2 and res[1]["rc"] == 2, res                        # MCMCX_INTERRUPTED on both
    assert res[0]["simuind"] == res[1]["simuind"] and res[0]["simuind"] % 50 == 0 and 100 < res[0]["simuind"] < 2000000
    assert res[0]["W"] == res[1]["W"] and res[0]["R00"] == res[1]["R00"]       # the same pooled state on both ranks


def _workers(tmp_path, mode, signal_rank=None):
    import signal
    import time
    import uuid
    key = "w%s" % uuid.uuid4().hex[:12]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MCMCX_COMM_KEY")}
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multirank_worker.py"), key, str(r), "2", str(tmp_path), mode],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(2)]
    try:
        if signal_rank is not None:
            t0 = time.time()
            while not all((tmp_path / ("rank%d.ready" % r)).exists() for r in range(2)):
                assert time.time() - t0 < 240 and all(p.poll() is None for p in procs), [p.stdout.read().decode()[-2000:] for p in procs if p.poll() is not None]
                time.sleep(0.05)
            time.sleep(0.5)
            procs[signal_rank].send_signal(signal.SIGUSR1)
        outs = [p.communicate(timeout=400)[0].decode(errors="replace") for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), outs
    return [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]


def test_a_signal_past_the_last_tick_stops_the_rank_that_caught_it(tmp_path):
    """with no tick ahead (here adaptend = 100) no collective is left in which a peer could be stranded, so the
    rank that caught the signal leaves at its next launch boundary by itself instead of ignoring the signal until `upto`;
    its peer, which saw no signal, runs to the end."""
    res = _workers(tmp_path, "tail", signal_rank=1)
    assert res[1]["rc"] == 2 and 100 < res[1]["simuind"] < 1200000, res
    assert res[0]["rc"] == 0 and res[0]["simuind"] == 1200000, res


def test_a_run_resumed_after_a_collective_stop_equals_an_uninterrupted_one(tmp_path):
    """the tick at which the ranks agree to stop is applied before they return, so clearing the flag and
    calling mcmcx_run again continues the trajectory of a run that was never interrupted, bit for bit."""
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    res = _workers(tmp_path / "a", "resume")
    ref = _workers(tmp_path / "b", "plain")
    assert res[0]["stops"] == [150] and res[1]["stops"] == [150], res      # the first tick of mcmcx_run(101..): both ranks, same place
    assert ref[0]["stops"] == [] and ref[1]["stops"] == []
    for r in range(2):
        assert res[r]["rc"] == 0 and res[r]["simuind"] == 1000
        for k in ("theta_bits", "R_bits", "W", "R00", "theta_sum"):
            assert res[r][k] == ref[r][k], (r, k, res[r], ref[r])
