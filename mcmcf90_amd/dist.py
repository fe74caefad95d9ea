"""Host-side arithmetic around the pooled moment vector [count, sum (d), upper second moments (d(d+1)/2)] of the
multi-GPU form of the path: how chains are sharded over ranks (the rule bench.py and the Fortran shim's `ngpus` follow)
and how mean / covariance come out of the vector.  The exchange itself is in libmcmcx.so (csrc/mcx_comm.hpp:
mcmcx_comm_create / mcmcx_allreduce_moments, RCCL); nothing here touches a device."""
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
