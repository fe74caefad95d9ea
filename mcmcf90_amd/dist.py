"""Host-side helpers for the multi-GPU form of the path: how chains are sharded over ranks and how the
pooled empirical moments are combined.  No device code here: the per-GPU partial comes from
`Engine.pooled_moments[_dev]` (libmcmcx.so); the exchange is a torch.distributed all-reduce (RCCL on GPUs, gloo in the
CPU tests)."""
import numpy as np


def shard(nchains_total, rank, world):
    """Contiguous block of chains of a rank: (n_local, chain_id0).  chain_id0 is the Philox key word of the
    block's first chain, so every chain draws the same stream whatever the GPU count."""
    if nchains_total % world != 0:
        raise ValueError("nchains_total must be divisible by the world size")
    n = nchains_total // world
    return n, rank * n


def moments_len(d):
    return 1 + d + d * (d + 1) // 2


def allreduce_moments(vec, dist=None):
    """Sum the per-rank pooled moment vectors [count, sum (d), second moments (d(d+1)/2)].  `vec` is a torch
    tensor (on the GPU for RCCL, on the CPU for gloo); in place."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(vec)
    return vec


def finalize_moments(vec, d, shift):
    """Pooled mean and covariance from [count, sum_c x, sum_c x x'] of x = theta - shift."""
    v = np.asarray(vec, dtype=np.float64)
    n = v[0]
    s1 = v[1:1 + d]
    m = s1 / n
    s2 = np.zeros((d, d))
    k = 1 + d
    for j in range(d):
        for i in range(j + 1):
            s2[i, j] = s2[j, i] = v[k + j * (j + 1) // 2 + i]
    cov = (s2 - n * np.outer(m, m)) / (n - 1.0)
    return np.asarray(shift) + m, cov
