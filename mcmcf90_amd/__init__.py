"""mcmcf90_amd -- MI355X-native adaptive-Metropolis engine behind mcmcf90's user surface.

csrc/      HIP kernels + the C ABI (libmcmcx.so, include/mcmcx.h)
engine.py  Python mirror of the reference's driver-program surface (tests, bench)
fortran/   ISO_C_BINDING shim: module mcmcmod + mcmc_main for existing Fortran drivers
"""
from .engine import Engine, Comm, McmcError, make_config, engine_from_problem  # noqa: F401
