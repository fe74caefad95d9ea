"""Build the HIP engine in-tree: mcmcf90_amd/libmcmcx.so (gfx950 only).

    python -m mcmcf90_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", f) for f in ("mcx_api.hip", "mcx_kernels.hpp", "mcx_group.hpp", "mcx_group_ram.hpp", "mcx_pooled2.hpp", "mcx_device.hpp", "mcx_comm.hpp")]
HDR = os.path.join(os.path.dirname(HERE), "include", "mcmcx.h")
LIB = os.path.join(HERE, "libmcmcx.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the kernels spell out every fma themselves
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-Wno-unused-value", "-Wno-cuda-compat"]


def source_sha():
    """sha of everything that decides which instructions run: kernels, device functions, host orchestration (kernel choice,
    LDS sizes, launch geometry) and the communicator.  bench.py and tools/ key the stored PMC figures on it."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SRC):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in SRC + [HDR])


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    cmd = [HIPCC] + FLAGS + ["-o", LIB, SRC[0], "-L/opt/rocm/lib", "-lrccl", "-lrt", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
