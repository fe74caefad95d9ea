"""The sharding rule and the moment-vector arithmetic on CPU: two gloo ranks take their blocks of chains, sum their pooled
moment vectors and finalize the same mean/covariance a single process gets from all chains.  (The product's own
exchange -- libmcmcx.so's communicator -- is covered by tests/test_comm_cpu.py without a GPU and by
tests/test_gpu_multirank.py with one.)"""
import os
import sys
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _local_moments(x):
    n, d = x.shape
    out = np.zeros(1 + d + d * (d + 1) // 2)
    out[0] = n
    out[1:1 + d] = x.sum(axis=0)
    k = 1 + d
    for j in range(d):
        for i in range(j + 1):
            out[k + j * (j + 1) // 2 + i] = np.dot(x[:, i], x[:, j])
    return out


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from mcmcf90_amd import dist as mdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d, ntot = 6, 512
    rng = np.random.default_rng(123)
    theta = rng.standard_normal((ntot, d)) @ rng.standard_normal((d, d)) + 3.0
    shift = np.full(d, 3.0)
    n, c0 = mdist.shard(ntot, rank, world)
    assert (n, c0) == (ntot // world, rank * (ntot // world))
    v = torch.from_numpy(_local_moments(theta[c0:c0 + n] - shift))
    dist.all_reduce(v)                                     # (on a GPU node: mcmcx_allreduce_moments, RCCL, inside libmcmcx.so)
    mean, cov = mdist.finalize_moments(v.numpy(), d, shift)
    if rank == 0:
        q.put((mean, cov, theta))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_pooled_moments_match_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    mean, cov, theta = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    np.testing.assert_allclose(mean, theta.mean(axis=0), rtol=1e-12)
    np.testing.assert_allclose(cov, np.cov(theta.T), rtol=1e-9, atol=1e-12)


def test_shard_is_a_partition():
    sys.path.insert(0, ROOT)
    from mcmcf90_amd import dist as mdist
    ids = []
    for r in range(8):
        n, c0 = mdist.shard(1048576, r, 8)
        ids.append((c0, c0 + n))
    assert ids[0][0] == 0 and ids[-1][1] == 1048576
    assert all(ids[i][1] == ids[i + 1][0] for i in range(7))
    assert mdist.moments_len(50) == 1326
