"""ctypes binding of libmcmcx.so (the C ABI declared in include/mcmcx.h).

There is no CPU fallback: if the shared library is missing, or no HIP device is present,
the engine raises.  The library is never built implicitly at import time on a GPU box --
__graft_entry__.build() / `python -m mcmcf90_amd.build` does that.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.environ.get("MCMCX_LIBRARY") or os.path.join(HERE, "libmcmcx.so")      # override: profiling / debug builds

_DP = C.POINTER(C.c_double)
_IP = C.POINTER(C.c_int32)


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("npar", "nchains", "method", "nsimu", "doadapt", "doburnin", "adaptint", "adapthist", "badaptint",
                 "adaptend", "initcmatn", "burnintime", "greedy", "updatesigma")] + \
               [(n, C.c_double) for n in
                ("scalelimit", "scalefactor", "drscale", "N0", "S02", "condmax", "alphatarget", "nuparam")] + \
               [("seed", C.c_uint32), ("chain_id0", C.c_uint32), ("record_accept", C.c_int32),
                ("record_chain", C.c_int32), ("device", C.c_int32), ("pooled", C.c_int32), ("scam_fast", C.c_int32)]


# name -> (restype, argtypes); every symbol include/mcmcx.h declares
SYMBOLS = {
    "mcmcx_config_defaults": (None, [C.POINTER(Config)]),
    "mcmcx_create": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "mcmcx_destroy": (C.c_int, [C.c_void_p]),
    "mcmcx_last_error": (C.c_char_p, []),
    "mcmcx_version": (C.c_char_p, []),
    "mcmcx_device_count": (C.c_int32, []),
    "mcmcx_device_info": (C.c_int, [C.c_int32, C.c_char_p, C.c_int32]),
    "mcmcx_device_ident": (C.c_int, [C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_int32)]),
    "mcmcx_last_kernel": (C.c_char_p, [C.c_void_p]),
    "mcmcx_debug_kernel_table": (C.c_int, [C.c_int32, C.c_char_p, C.c_int32]),
    "mcmcx_set_par0": (C.c_int, [C.c_void_p, _DP, C.c_int32]),
    "mcmcx_set_cmat0": (C.c_int, [C.c_void_p, _DP, C.c_int32]),
    "mcmcx_set_sigma2nobs": (C.c_int, [C.c_void_p, _DP, _IP, C.c_int32]),
    "mcmcx_set_target_gauss": (C.c_int, [C.c_void_p, _DP, _DP]),
    "mcmcx_set_target_banana": (C.c_int, [C.c_void_p, C.c_double]),
    "mcmcx_set_target_expdata": (C.c_int, [C.c_void_p, C.c_int32, _DP, _DP]),
    "mcmcx_set_target_expdata_cols": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _DP, _DP]),
    "mcmcx_set_bounds": (C.c_int, [C.c_void_p, _DP, _DP]),
    "mcmcx_set_priors": (C.c_int, [C.c_void_p, _DP, _DP]),
    "mcmcx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mcmcx_init": (C.c_int, [C.c_void_p]),
    "mcmcx_run": (C.c_int, [C.c_void_p, C.c_int32]),
    "mcmcx_sync": (C.c_int, [C.c_void_p]),
    "mcmcx_simuind": (C.c_int32, [C.c_void_p]),
    "mcmcx_get_counters": (C.c_int, [C.c_void_p, C.c_int32, _IP]),
    "mcmcx_get_totals": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "mcmcx_get_totals_n": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int32]),
    "mcmcx_get_theta": (C.c_int, [C.c_void_p, _DP]),
    "mcmcx_get_scalars": (C.c_int, [C.c_void_p, _DP]),
    "mcmcx_get_rng": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_uint64), _IP, _DP]),
    "mcmcx_get_R": (C.c_int, [C.c_void_p, C.c_int32, _DP]),
    "mcmcx_get_qcovstd": (C.c_int, [C.c_void_p, C.c_int32, _DP]),
    "mcmcx_get_dr": (C.c_int, [C.c_void_p, C.c_int32, _DP, _DP]),
    "mcmcx_get_chaincov": (C.c_int, [C.c_void_p, C.c_int32, _DP, _DP, _DP]),
    "mcmcx_get_accepted": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_uint8)]),
    "mcmcx_get_accept_masks": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), _IP]),
    "mcmcx_get_chain": (C.c_int, [C.c_void_p, C.c_int32, _DP, _DP, _DP, _IP]),
    "mcmcx_pooled_moments": (C.c_int, [C.c_void_p, _DP]),
    "mcmcx_pooled_moments_dev": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mcmcx_pooled_moments_len": (C.c_int32, [C.c_void_p]),
    "mcmcx_debug_math": (C.c_int, [C.c_int32, C.c_int32, _DP, _DP, _DP]),
    "mcmcx_debug_rng": (C.c_int, [C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_double, C.c_double, _DP,
                                  C.POINTER(C.c_uint64)]),
    "mcmcx_debug_set_factor": (C.c_int, [C.c_void_p, _DP, _DP]),
    "mcmcx_kernel_time": (C.c_int, [C.c_void_p, _DP, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int]),
}

SSFUN_T = C.CFUNCTYPE(None, _DP, C.c_int32, C.c_int32, _DP, C.c_void_p)
PRIORFUN_T = C.CFUNCTYPE(C.c_double, _DP, C.c_int32, C.c_void_p)
CHECKBOUNDS_T = C.CFUNCTYPE(C.c_int32, _DP, C.c_int32, C.c_void_p)
EXCHANGE_T = C.CFUNCTYPE(None, C.c_void_p)
SYMBOLS["mcmcx_set_exchange"] = (C.c_int, [C.c_void_p, EXCHANGE_T, C.c_void_p, C.c_void_p])
SYMBOLS["mcmcx_get_pooled"] = (C.c_int, [C.c_void_p, _DP, _DP, _DP, _DP])
SSFUN_ER_T = C.CFUNCTYPE(None, _DP, C.c_int32, C.c_int32, C.c_double, _DP, C.c_void_p)
SYMBOLS["mcmcx_set_target_host_er"] = (C.c_int, [C.c_void_p, SSFUN_ER_T])
SYMBOLS["mcmcx_install_signal_handlers"] = (C.c_int, [])
SYMBOLS["mcmcx_interrupted"] = (C.c_int, [])
SYMBOLS["mcmcx_clear_interrupt"] = (None, [])
SYMBOLS["mcmcx_set_target_host"] = (C.c_int, [C.c_void_p, SSFUN_T, PRIORFUN_T, CHECKBOUNDS_T, C.c_void_p])

SSFUN_BATCH_T = C.CFUNCTYPE(None, _DP, C.c_int32, C.c_int32, C.c_int32, _DP, C.c_void_p)
SYMBOLS["mcmcx_set_target_host_batch"] = (C.c_int, [C.c_void_p, SSFUN_BATCH_T, PRIORFUN_T, CHECKBOUNDS_T, C.c_void_p, C.c_int32])
SYMBOLS["mcmcx_set_target_module"] = (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_int64])
_HP = C.POINTER(C.c_void_p)
SYMBOLS["mcmcx_comm_create"] = (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _HP])
SYMBOLS["mcmcx_comm_create_all"] = (C.c_int, [C.c_int32, _IP, _HP])
SYMBOLS["mcmcx_comm_destroy"] = (C.c_int, [C.c_void_p])
SYMBOLS["mcmcx_comm_rank"] = (C.c_int32, [C.c_void_p])
SYMBOLS["mcmcx_comm_size"] = (C.c_int32, [C.c_void_p])
SYMBOLS["mcmcx_comm_barrier"] = (C.c_int, [C.c_void_p])
SYMBOLS["mcmcx_comm_allreduce_host"] = (C.c_int, [C.c_void_p, _DP, C.c_int32, C.c_int32])
SYMBOLS["mcmcx_set_comm"] = (C.c_int, [C.c_void_p, C.c_void_p])
SYMBOLS["mcmcx_allreduce_moments"] = (C.c_int, [C.c_void_p, _DP])
SYMBOLS["mcmcx_allreduce_moments_all"] = (C.c_int, [_HP, C.c_int32, _DP])
SYMBOLS["mcmcx_run_all"] = (C.c_int, [_HP, C.c_int32, C.c_int32])
# MCMC_run1 / MCMC_run1_er: the arithmetic of one invocation (the caller evaluates the target)
SYMBOLS["mcmcx_set_target_external"] = (C.c_int, [C.c_void_p])
SYMBOLS["mcmcx_run1_decide"] = (C.c_int, [C.c_void_p, C.c_int32, _DP, _DP, _DP, _DP, _DP, _DP, _DP, _DP, _DP, _DP, _DP, _IP])
SYMBOLS["mcmcx_run1_propose"] = (C.c_int, [C.c_void_p, C.c_int32, _DP, _DP])
SYMBOLS["mcmcx_run1_sscrit"] = (C.c_int, [C.c_void_p, _DP, _DP, _DP])

_lib = None


def load():
    """dlopen libmcmcx.so and bind every declared symbol (raises if the library or a symbol is missing)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIBPATH):
            raise RuntimeError("mcmcf90_amd: %s is missing -- build it with `python -m mcmcf90_amd.build` "
                               "(there is no CPU fallback)" % LIBPATH)
        L = C.CDLL(LIBPATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)          # AttributeError if the symbol is not exported
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib
