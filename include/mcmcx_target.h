/*
 * include/mcmcx_target.h -- device-side header for a USER TARGET MODULE of the mcmcx engine.
 *
 * The reference calls the user's ssfunction / priorfun / checkbounds (external_inc.h:4-33) once per proposal on the
 * host.  Through mcmcx_set_target_host the engine does the same (PCIe round trip per stage).  A target module is the
 * GPU-speed form of that surface: the user writes the three functions as HIP __device__ code, this header wraps them
 * into one kernel, `hipcc --genco --offload-arch=gfx950 -ffp-contract=off my_target.hip -o my_target.hsaco` compiles
 * it, and mcmcx_set_target_module(handle, "my_target.hsaco", "my_target", data, nbytes) loads it (hipModuleLoad).
 * The engine launches the kernel where the reference calls the functions (MCMC_run.F90:47,55-56,69,74-75), one
 * thread per chain, candidates and results staying in HBM:
 *
 *     #include "mcmcx_target.h"
 *     __device__ void   my_ss(const double *theta, int npar, int ny, const void *data, double *ss) { ... ss[0] = ...; }
 *     __device__ double my_prior(const double *theta, int npar, const void *data) { return 0.0; }
 *     __device__ int    my_bounds(const double *theta, int npar, const void *data) { return 1; }
 *     MCMCX_DEFINE_TARGET(my_target, my_ss, my_prior, my_bounds)
 *
 * `data` points at a device copy of the bytes given to mcmcx_set_target_module (observations, constants).  theta is a
 * private copy of the chain's candidate (npar <= MCMCX_TARGET_MAX_NPAR; define it before the include to change it).
 * Order per chain, as in the reference: checkbounds; if inside, priorfun, then ssfunction.  With method = 'er' the
 * prior and the sum of squares are asked for separately (MCMC_run_er.F90:54-76) and ssfunction always sums to the
 * end, like the library's default ssfunction_er (ssfunction_er0.f90: "no er for ss").
 */
#ifndef MCMCX_TARGET_H
#define MCMCX_TARGET_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef MCMCX_TARGET_MAX_NPAR
#define MCMCX_TARGET_MAX_NPAR 64
#endif
#ifndef MCMCX_TARGET_MAX_NY
#define MCMCX_TARGET_MAX_NY 8                  /* response columns of ssfunction (nycol); define it before the include to change it */
#endif
#define MCMCX_TARGET_ABI 1
#define MCMCX_HE_INB 0        /* result slots of a chain: in bounds (1.0 / 0.0), prior, ss per response column */
#define MCMCX_HE_PRI 1
#define MCMCX_HE_SS 2
#define MCMCX_HX_STAGE2 3     /* engine state between the phases: != 0 -> this chain asks for the delayed-rejection / er evaluation */
#define MCMCX_HX_CRIT 6       /* method = 'er': the threshold sscrit of this chain */

/* the kernel's one argument (passed by value); every per-chain array is tile-interleaved:
 * element k of chain c sits at base[((c / 64) * K + k) * 64 + c % 64] */
typedef struct mcmcx_target_args {
    const double *src;        /* candidates, K = stride_k */
    double *hev;              /* results, K = nhe = 2 + ny */
    const double *hx;         /* K = nhx */
    const void *userdata;
    int32_t stride_k, npar, ny, nhe, nhx, nchains;
    int32_t use_stage2;       /* 1: only chains with hx[MCMCX_HX_STAGE2] != 0 are evaluated */
    int32_t what;             /* 0: bounds + prior + ss; 1: bounds + prior only; 2: ss only, threshold in hx[MCMCX_HX_CRIT] */
} mcmcx_target_args;

#define MCMCX_TI(base, K, k, c) (base)[(((size_t)((c) >> 6)) * (size_t)(K) + (size_t)(k)) * 64 + ((c) & 63)]

#define MCMCX_DEFINE_TARGET(NAME, SSFUN, PRIORFUN, BOUNDSFUN)                                                          \
    extern "C" __device__ const int NAME##_abi = MCMCX_TARGET_ABI;                                                       \
    extern "C" __device__ const int NAME##_max_npar = MCMCX_TARGET_MAX_NPAR;                                             \
    extern "C" __device__ const int NAME##_max_ny = MCMCX_TARGET_MAX_NY;                                                 \
    extern "C" __global__ void __launch_bounds__(64) NAME(mcmcx_target_args a)                                           \
    {                                                                                                                    \
        const int c = blockIdx.x * 64 + threadIdx.x;                                                                     \
        double th[MCMCX_TARGET_MAX_NPAR], ss[MCMCX_TARGET_MAX_NY];                                                       \
        for (int j = 0; j < MCMCX_TARGET_MAX_NY; ++j) ss[j] = 0.0;                                                       \
        const bool skip = (c >= a.nchains) || (a.use_stage2 && MCMCX_TI(a.hx, a.nhx, MCMCX_HX_STAGE2, c) == 0.0);        \
        int inb = 1;                                                                                                     \
        double pri = 0.0;                                                                                                \
        if (!skip) {                                                                                                     \
            for (int k = 0; k < a.npar; ++k) th[k] = MCMCX_TI(a.src, a.stride_k, k, c);                                  \
            if (a.what != 2) {                                                                                           \
                inb = BOUNDSFUN(th, a.npar, a.userdata) ? 1 : 0;                                                         \
                if (inb) pri = PRIORFUN(th, a.npar, a.userdata);                                                         \
            }                                                                                                            \
            if ((a.what == 0 && inb) || a.what == 2) SSFUN(th, a.npar, a.ny, a.userdata, ss);                            \
        }                                                                                                                \
        MCMCX_TI(a.hev, a.nhe, MCMCX_HE_INB, c) = (inb && !skip) ? 1.0 : 0.0;                                            \
        MCMCX_TI(a.hev, a.nhe, MCMCX_HE_PRI, c) = pri;                                                                   \
        for (int j = 0; j < a.ny; ++j) MCMCX_TI(a.hev, a.nhe, MCMCX_HE_SS + j, c) = ss[j];                               \
    }

#endif
