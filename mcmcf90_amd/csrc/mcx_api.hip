// mcx_api.hip -- host side of libmcmcx.so: the C ABI of include/mcmcx.h.
//
// Owns the device state of N chains, cuts the iteration range into launches of the
// step kernel separated by the adaptation ticks of MCMC_adapt (MCMC_adapt.F90:40-46),
// and decodes the device history back into the reference's chain/sschain/s2chain
// form.  No CPU fallback: every numerical step of the sampler runs in the HIP kernels
// of mcx_kernels.hpp; without a GPU every entry point that needs one fails.
// ONE translation unit: the host parts below are included in order (engine object, initial factor, kernel selection and
// launchers, adaptation tick, pooled mode and communicator, host callbacks); this file holds the extern "C" entry points.
#include <hip/hip_runtime.h>
#include <csignal>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include "../../include/mcmcx.h"
#include "../../include/mcmcx_target.h"
#include "mcx_kernels.hpp"
#include "mcx_group.hpp"
#include "mcx_group_ram.hpp"

#include "mcx_host_engine.hpp"
#include "mcx_host_linalg.hpp"
#include "mcx_host_launch.hpp"
#include "mcx_host_adapt.hpp"
#include "mcx_host_pooled.hpp"
#include "mcx_host_callbacks.hpp"

// ------------------------------------------------------------------ C ABI
extern "C" {

const char *mcmcx_last_error(void) { return g_err.c_str(); }
const char *mcmcx_version(void) { return "mcmcx 0.3 (gfx950)"; }
int32_t mcmcx_device_count(void) { int n = 0; return (hipGetDeviceCount(&n) == hipSuccess) ? n : 0; }
int mcmcx_device_info(int32_t device, char *buf, int32_t len)
{
    if (!buf || len < 2) return fail(-1, "mcmcx_device_info: bad argument");
    hipDeviceProp_t prop;
    char bus[64] = "?";
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { snprintf(buf, (size_t)len, "device %d: %s", device, hipGetErrorString(e)); return fail(-10, buf); }
    (void)hipDeviceGetPCIBusId(bus, (int)sizeof bus, device);
    snprintf(buf, (size_t)len, "%s %s, pci %s, %d CUs, %.0f GiB", prop.name, prop.gcnArchName, bus, prop.multiProcessorCount,
        (double)prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0));
    return 0;
}

// what identifies the physical device and its clock limits (bench.py puts it in its JSON line: the pool's boxes differ by ~8 %)
int mcmcx_device_ident(int32_t device, char *uuid_hex, int32_t len, int32_t *out5)
{
    if (!uuid_hex || len < 33 || !out5) return fail(-1, "mcmcx_device_ident: bad argument (uuid buffer of >= 33 bytes, five integers)");
    hipUUID u;
    HIPCHK(hipDeviceGetUuid(&u, device));
    for (int i = 0; i < 16; ++i) snprintf(uuid_hex + 2 * i, 3, "%02x", (unsigned)(unsigned char)u.bytes[i]);
    int v = 0;
    HIPCHK(hipDeviceGetAttribute(&v, hipDeviceAttributeClockRate, device)); out5[0] = v;                 // kHz, the engine clock's limit
    HIPCHK(hipDeviceGetAttribute(&v, hipDeviceAttributeMemoryClockRate, device)); out5[1] = v;           // kHz
    HIPCHK(hipDeviceGetAttribute(&v, hipDeviceAttributeMemoryBusWidth, device)); out5[2] = v;            // bits
    HIPCHK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device)); out5[3] = v;
    HIPCHK(hipDeviceGetAttribute(&v, hipDeviceAttributeL2CacheSize, device)); out5[4] = v;
    return 0;
}

const char *mcmcx_last_kernel(mcmcx_handle h)
{
    static thread_local std::string name;
    name = h ? h->last_kernel : "";
    // template instances are launched as (k<...>)
    if (name.size() >= 2 && name.front() == '(' && name.back() == ')') name = name.substr(1, name.size() - 2);
    return name.c_str();
}

void mcmcx_config_defaults(mcmcx_config *c)                      // mcmcinit.F90:184-230
{
    memset(c, 0, sizeof *c);
    c->npar = 0; c->nchains = 1; c->method = MCMCX_METHOD_DRAM;
    c->nsimu = 0; c->doadapt = 1; c->doburnin = 0; c->burnintime = 0; c->badaptint = -1;
    c->greedy = 0; c->scalelimit = 0.05; c->scalefactor = 2.5; c->drscale = 0.0;
    c->adaptint = 100; c->adapthist = 0; c->adaptend = 0; c->initcmatn = 0;
    c->N0 = 1.0; c->S02 = 0.0; c->updatesigma = 1; c->condmax = 0.0;
    c->alphatarget = 0.234; c->nuparam = 0.7;
    c->seed = MCMCX_DEFAULT_SEED; c->chain_id0 = 0; c->record_accept = 0; c->record_chain = 0; c->device = 0; c->scam_fast = 0;
}

int mcmcx_create(const mcmcx_config *cfg_in, mcmcx_handle *out)
{
    if (!cfg_in || !out) return fail(-1, "mcmcx_create: null argument");
    mcmcx_config c = *cfg_in;
    // check_mcmcinit_parameters, mcmcinit.F90:235-368
    if (c.adapthist < 0) c.adapthist = 0;
    if (c.adaptint < 0) { c.adaptint = 0; c.doadapt = 0; }
    if (c.burnintime < 0) c.burnintime = 0;
    if (c.badaptint <= 0) c.badaptint = c.adaptint;
    if (c.badaptint == 0) c.doburnin = 0;
    if (c.initcmatn < 0) c.initcmatn = 0;
    if (c.scalelimit < 0.0 || c.scalelimit > 0.5)
        return fail(-2, "ERROR: Scalelimit control variable should be between [0,0.5]");
    if (c.scalefactor < 0.0) c.scalefactor = 1.0;
    if (c.method != MCMCX_METHOD_DRAM && c.method != MCMCX_METHOD_RAM && c.method != MCMCX_METHOD_SCAM &&
        c.method != MCMCX_METHOD_ER) return fail(-3, "unknown method");
    if (c.method == MCMCX_METHOD_ER) c.drscale = 0.0;                                // 'no dr with er', MCMC_run_er.F90:26-29
    if (c.method == MCMCX_METHOD_SCAM) {                                             // mcmcinit.F90:321-330
        if (c.condmax <= 0.0) c.condmax = 1.0e15;
        c.doburnin = 0; c.drscale = 0.0;
    }
    if (c.method == MCMCX_METHOD_RAM) c.drscale = 0.0;
    if (c.method != MCMCX_METHOD_SCAM) c.scam_fast = 0;
    if (c.nsimu < 1) return fail(-4, "nsimu <= 0 stopping");                         // mcmc_main.F90:22-25
    if (c.npar < 1 || c.npar > MCX_MAX_NPAR) return fail(-5, "npar must be in 1.." + std::to_string(MCX_MAX_NPAR));
    if (c.nchains < 1) return fail(-5, "nchains must be >= 1");
    if (c.doadapt && c.method != MCMCX_METHOD_RAM) {
        if (c.adaptint == 0) return fail(-7, "doadapt with adaptint = 0");
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1) return fail(-10, "no HIP device: the mcmcx engine has no CPU fallback");
    if (c.device < 0 || c.device >= ndev) return fail(-10, "bad device ordinal");
    HIPCHK(hipSetDevice(c.device));
    mcmcx_engine *h = new mcmcx_engine();
    h->sw.read();
    h->cfg = c; h->d = c.npar; h->P = c.npar * (c.npar + 1) / 2;
    h->ntiles = (c.nchains + 63) / 64; h->nlanes = h->ntiles * 64;
    h->dodr = (c.drscale > 0.0) ? 1 : 0; h->usesvd = (c.condmax > 0.0) ? 1 : 0; h->pooled = c.pooled ? 1 : 0;
    e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail(-100, hipGetErrorString(e)); }
    h->own_stream = true;
    *out = h;
    return 0;
}

int mcmcx_destroy(mcmcx_handle h)
{
    if (!h) return 0;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto &p : h->pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (void *p : h->allocs) (void)hipFree(p);
    for (void *p : h->hallocs) (void)hipHostFree(p);
    h->h_cand.release(); h->h_ev.release(); h->h_hx.release();
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    if (h->mod) (void)hipModuleUnload(h->mod);
    delete h;
    return 0;
}

int mcmcx_set_stream(mcmcx_handle h, void *s)
{
    if (!h) return fail(-1, "null handle");
    if (h->inited) return fail(-1, "mcmcx_set_stream after mcmcx_init");
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)s; h->own_stream = false;
    return 0;
}

int mcmcx_set_par0(mcmcx_handle h, const double *p, int32_t n)
{
    if (!h || !p) return fail(-1, "null argument");
    if (n != h->d) return fail(-20, "par0 and npar dont match");                     // MCMC_init.F90:173
    h->par0.assign(p, p + n);
    return 0;
}
int mcmcx_set_cmat0(mcmcx_handle h, const double *c, int32_t n)
{
    if (!h || !c) return fail(-1, "null argument");
    if (n != h->d) return fail(-20, "cmat0 and npar dont match");                    // MCMC_init.F90:207
    h->cmat0.assign(c, c + (size_t)n * n);
    return 0;
}
int mcmcx_set_sigma2nobs(mcmcx_handle h, const double *s2, const int32_t *nobs, int32_t nycol)
{
    if (!h || !s2 || !nobs) return fail(-1, "null argument");
    if (nycol < 1 || nycol > MCX_MAX_NYCOL) return fail(-21, "nycol must be in 1.." + std::to_string(MCX_MAX_NYCOL));
    if (h->inited) return fail(-20, "set sigma2 / nobs before mcmcx_init");
    h->ny = nycol; h->sigma2v.assign(s2, s2 + nycol); h->nobsv.assign(nobs, nobs + nycol);
    h->sigma2 = s2[0]; h->nobs = nobs[0]; h->sigma2ok = true;
    return 0;
}
int mcmcx_set_target_gauss(mcmcx_handle h, const double *mu, const double *lam)
{
    if (!h || !mu || !lam) return fail(-1, "null argument");
    h->tkind = TGT_GAUSS; h->tmu.assign(mu, mu + h->d); h->tlam.assign(lam, lam + (size_t)h->d * h->d);
    return 0;
}
int mcmcx_set_target_banana(mcmcx_handle h, double b)
{
    if (!h) return fail(-1, "null handle");
    if (h->d < 2) return fail(-22, "banana target needs npar >= 2");
    h->tkind = TGT_BANANA; h->tb = b;
    return 0;
}
int mcmcx_set_target_expdata(mcmcx_handle h, int32_t n, const double *x, const double *y)
{
    if (!h || !x || !y || n < 1) return fail(-1, "bad argument");
    if (h->d != 2) return fail(-22, "expdata target needs npar = 2");
    h->tkind = TGT_EXPDATA; h->tx.assign(x, x + n); h->ty.assign(y, y + n);
    return 0;
}
int mcmcx_set_target_expdata_cols(mcmcx_handle h, int32_t n, int32_t nycol, const double *x, const double *y)
{
    if (!h || !x || !y || n < 1) return fail(-1, "bad argument");
    if (nycol < 1 || nycol > MCX_MAX_NYCOL) return fail(-21, "nycol must be in 1.." + std::to_string(MCX_MAX_NYCOL));
    if (h->d != 1 + nycol) return fail(-22, "response-column target needs npar = 1 + nycol");
    h->tkind = TGT_EXPCOLS; h->tncols = nycol; h->tx.assign(x, x + n); h->ty.assign(y, y + (size_t)n * nycol);
    return 0;
}
int mcmcx_set_target_host(mcmcx_handle h, mcmcx_ssfun_t ss, mcmcx_priorfun_t pri, mcmcx_checkbounds_t cb, void *user)
{
    if (!h || !ss) return fail(-1, "mcmcx_set_target_host: ssfunction is required");     // ssfunction0.f90:10-14
    h->tkind = TGT_HOST; h->h_ss = ss; h->h_ss_batch = nullptr; h->h_pri = pri; h->h_cb = cb; h->h_user = user;
    return 0;
}

int mcmcx_set_target_host_batch(mcmcx_handle h, mcmcx_ssfun_batch_t ss_batch, mcmcx_priorfun_t pri, mcmcx_checkbounds_t cb, void *user,
    int32_t nthreads)
{
    if (!h || !ss_batch) return fail(-1, "mcmcx_set_target_host_batch: ssfunction_batch is required");
    if (h->inited) return fail(-20, "set the target before mcmcx_init");
    h->tkind = TGT_HOST; h->h_ss_batch = ss_batch; h->h_ss = nullptr; h->h_pri = pri; h->h_cb = cb; h->h_user = user;
    h->h_threads = nthreads < 1 ? 1 : nthreads;
    return 0;
}

int mcmcx_set_target_external(mcmcx_handle h)
{
    if (!h) return fail(-1, "null handle");
    if (h->inited) return fail(-2, "mcmcx_set_target_external after mcmcx_init");
    h->tkind = TGT_HOST; h->h_ss = nullptr; h->h_ss_batch = nullptr; h->h_pri = nullptr; h->h_cb = nullptr; h->h_user = nullptr;
    h->external = true;
    return 0;
}

int mcmcx_set_target_module(mcmcx_handle h, const char *code_object_path, const char *kernel_name, const void *userdata, int64_t nbytes)
{
    if (!h || !code_object_path || !kernel_name) return fail(-1, "mcmcx_set_target_module: null argument");
    if (h->inited) return fail(-20, "set the target before mcmcx_init");
    HIPCHK(hipSetDevice(h->cfg.device));
    hipError_t e = hipModuleLoad(&h->mod, code_object_path);
    if (e != hipSuccess) return fail(-37, std::string("cannot load the target module ") + code_object_path + ": " + hipGetErrorString(e));
    e = hipModuleGetFunction(&h->mod_fn, h->mod, kernel_name);
    if (e != hipSuccess) return fail(-37,
        std::string("target module: no kernel named ") + kernel_name + " (define it with MCMCX_DEFINE_TARGET, include/mcmcx_target.h)");
    for (const char *suffix : {"_abi", "_max_npar"}) {
        hipDeviceptr_t p = nullptr; size_t sz = 0; int v = 0;
        e = hipModuleGetGlobal(&p, &sz, h->mod, (std::string(kernel_name) + suffix).c_str());
        if (e != hipSuccess || sz != sizeof(int)) return fail(-37,
            std::string("target module: ") + kernel_name + suffix + " is missing (not built with MCMCX_DEFINE_TARGET?)");
        HIPCHK(hipMemcpy(&v, p, sizeof(int), hipMemcpyDeviceToHost));
        if (suffix[1] == 'a' && v != MCMCX_TARGET_ABI) return fail(-37, "target module: built against another mcmcx_target.h (abi " +
            std::to_string(v) + ")");
        if (suffix[1] == 'm' && h->d > v) return fail(-37, "target module: npar = " + std::to_string(h->d)
            + " but the module was compiled with MCMCX_TARGET_MAX_NPAR = " + std::to_string(v));
    }
    {   // response columns the module's kernel has room for (modules built before round 5 have no such symbol: eight)
        hipDeviceptr_t p = nullptr; size_t sz = 0; int v = 8;
        if (hipModuleGetGlobal(&p, &sz, h->mod, (std::string(kernel_name) + "_max_ny").c_str()) == hipSuccess && sz == sizeof(int))
            HIPCHK(hipMemcpy(&v, p, sizeof(int), hipMemcpyDeviceToHost));
        else (void)hipGetLastError();
        h->mod_max_ny = v;
    }
    if (userdata && nbytes > 0) {
        HIPCHK(hipMalloc(&h->d_moddata, (size_t)nbytes));
        h->allocs.push_back(h->d_moddata);
        HIPCHK(hipMemcpy(h->d_moddata, userdata, (size_t)nbytes, hipMemcpyHostToDevice));
    }
    h->tkind = TGT_MODULE;
    return 0;
}

int mcmcx_set_target_host_er(mcmcx_handle h, mcmcx_ssfun_er_t ss_er)
{
    if (!h) return fail(-1, "null handle");
    if (h->inited) return fail(-20, "set the target before mcmcx_init");
    h->h_ss_er = ss_er;
    return 0;
}

int mcmcx_set_bounds(mcmcx_handle h, const double *lo, const double *hi)
{
    if (!h) return fail(-1, "null handle");
    h->has_lo = lo != nullptr; h->has_hi = hi != nullptr;
    if (lo) h->tlo.assign(lo, lo + h->d);
    if (hi) h->thi.assign(hi, hi + h->d);
    return 0;
}
int mcmcx_set_priors(mcmcx_handle h, const double *mu, const double *sig)
{
    if (!h || !mu || !sig) return fail(-1, "null argument");
    h->has_pri = true; h->tpmu.assign(mu, mu + h->d); h->tpsig.assign(sig, sig + h->d);
    return 0;
}

int mcmcx_init(mcmcx_handle h)
{
    if (!h) return fail(-1, "null handle");
    if (h->inited) return fail(1, "Warning(mcmcinit): allready inited");            // MCMC_init.F90:24-27
    HIPCHK(hipSetDevice(h->cfg.device));
    h->sw.read();                                    // the A/B switches as the environment has them now: fixed for the engine's life
    const mcmcx_config &c = h->cfg;
    const int d = h->d, P = h->P, T = h->ntiles;
    if ((int)h->par0.size() != d) return fail(-30, "user initialization error: par0 not set");
    if (h->cmat0.empty()) {                                                           // MCMC_initcmat0: identity
        h->cmat0.assign((size_t)d * d, 0.0);
        for (int i = 0; i < d; ++i) h->cmat0[(size_t)i * d + i] = 1.0;
    }
    if (!h->sigma2ok) { h->sigma2 = 1.0; h->nobs = 1; h->ny = 1; }                    // MCMC_init.F90:52-59
    if (h->ny == 1) { h->sigma2v.assign(1, h->sigma2); h->nobsv.assign(1, h->nobs); }
    const int ny = h->ny;
    if (ny > 1 && !phased(h)) return
        fail(-36, "nycol > 1 needs the host-callback or the response-column target (the other built-in targets have one column)");
    if (h->tkind == TGT_MODULE && ny > h->mod_max_ny)
        return fail(-36, "target module: nycol = " + std::to_string(ny) + " but the module was compiled with MCMCX_TARGET_MAX_NY = " +
            std::to_string(h->mod_max_ny));
    if (h->tkind == TGT_EXPCOLS
        && h->tncols != ny) return fail(-36, "response-column target: mcmcx_set_sigma2nobs must give one sigma2 / nobs per column");
    if (ny > 1 && h->pooled && (!fused_cols(h) || c.method == MCMCX_METHOD_SCAM))
        return fail(-36, "nycol > 1 in pooled mode: the device-resident response-column target only, and not with method = 'scam'");
    if (h->tkind < 0) return fail(-31, "no target: the device engine needs mcmcx_set_target_*");
    std::vector<double> Rp, Cp, Rfull, qstd0;
    int info = host_initial_R(d, h->cmat0, Rp, Cp);
    if (h->usesvd) {                                                                  // Cp (packed cmat0) is still needed
        if (info != 0) { Rp.assign(P, 0.0); info = 0; }
        info = host_initial_svd(d, h->cmat0, c.condmax, c.method == MCMCX_METHOD_SCAM, Rfull, qstd0);
    }
    if (info != 0) return fail(-32, "could not factor the initial covariance");      // MCMC_init.F90:110
    h->S02eff = (c.S02 <= 0.0) ? h->sigma2 : c.S02;                                   // MCMC_init.F90:114-116
    double shape = c.N0 / 2.0 + (double)h->nobs / 2.0;

    EngineDev &E = h->E;
    E.d = d; E.P = P; E.ntiles = T;
    // pooled RAM adapts on the host side: the kernels see a plain Metropolis step
    E.method = (c.method == MCMCX_METHOD_RAM && !h->pooled) ? M_RAM : (c.method == MCMCX_METHOD_ER ? M_ER : M_DRAM);
    E.usesvd = h->usesvd; E.doscam = (c.method == MCMCX_METHOD_SCAM) ? 1 : 0; E.condmax = c.condmax;
    E.scam_fast = c.scam_fast ? 1 : 0;
    E.Rf = E.R2f = E.qstd = E.Gw = E.Vw = nullptr;
    E.greedy = c.greedy; E.adapthist = c.adapthist; E.initcmatn = (double)c.initcmatn;
    E.dodr = h->dodr; E.updatesigma = c.updatesigma; E.doadapt = c.doadapt; E.doburnin = c.doburnin; E.burnintime = c.burnintime;
    E.gam_shape = shape; E.N0S02 = c.N0 * h->S02eff;
    E.ny = ny; E.hs = d + ny; E.ssv = E.s2v = E.ss2v = nullptr; E.gshapev = nullptr;
    E.alphatarget = c.alphatarget; E.drscale = c.drscale; E.scalelimit = c.scalelimit; E.scalefactor = c.scalefactor;
    E.k0 = c.seed; E.chain_id0 = c.chain_id0;
    E.dr_lds = (h->dodr && dr_fits_lds(h)) ? 1 : 0;
    {   // plain AM / Metropolis / ER step kernel: state and scratch vectors in LDS (4 d x 512 bytes per wave) when that costs no
        // occupancy -- eight waves per CU still fit (npar <= 10), or all tiles are resident at once anyway (few chains)
        const char *ev = getenv("MCMCX_LDS_SCRATCH");                      // A/B switch: 0 = off
        const size_t per_wave = (size_t)4 * d * 64 * sizeof(double);
        const int per_cu = (int)((size_t)160 * 1024 / per_wave);
        hipDeviceProp_t prop;
        int cus = 256;
        if (hipGetDeviceProperties(&prop, c.device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        const bool fits = per_cu >= 8 || (per_cu >= 1 && (long long)T <= (long long)per_cu * cus);
        E.lds_scratch = (!h->pooled && !h->dodr && c.method != MCMCX_METHOD_RAM && c.method != MCMCX_METHOD_SCAM && fits && !(ev
            && atoi(ev) == 0)) ? 1 : 0;
        // ... and the packed factor with it (step_kernel_ldsr) where one column panel covers npar and state + factor of all tiles
        // are resident at once: (2 npar + npar (npar + 1) / 2) x 512 bytes per wave, 38 KiB at npar = 10 = four waves per CU
        const size_t per_wave_r = (size_t)(2 * d + P) * 64 * sizeof(double);
        const int per_cu_r = (int)((size_t)160 * 1024 / per_wave_r);
        const bool fits_r = per_cu_r >= 8 || (per_cu_r >= 1 && (long long)T <= (long long)per_cu_r * cus);
        // MCMCX_LDS_SCRATCH=1: the state only (A/B)
        if (E.lds_scratch && !h->usesvd && d <= TW && fits_r && !(ev && atoi(ev) == 1)) E.lds_scratch = 2;
        // method='ram' (per-chain factor, Cholesky form): the factor DCHUD / DCHDD rewrite every iteration, and their rotations, in LDS for
        // the launch where one column panel covers npar and all tiles are resident at once (step_kernel_ram_ldsr)
        if (!h->pooled && c.method == MCMCX_METHOD_RAM && !h->usesvd && d <= RW && fits_r && !(ev && atoi(ev) == 0)) E.lds_scratch = 3;
    }
    // target
    E.tgt.kind = phased(h) ? (int)TGT_HOST : h->tkind;   // the kernels know one phase-cut mode; who evaluates is the host's business
    E.tgt.b = h->tb; E.tgt.ndata = (int)h->tx.size(); E.tgt.ncols = h->tncols;
    E.tgt.mu = E.tgt.x = E.tgt.y = E.tgt.lo = E.tgt.hi = E.tgt.pmu = E.tgt.psig = nullptr;
    int rc;
    E.tgt.lamT = nullptr;
    if (h->tkind == TGT_GAUSS) {
        if ((rc = dev_upload(h, &E.tgt.mu, h->tmu))) return rc;
        std::vector<double> lt((size_t)(d + 4) * d + 64, 0.0);     // transpose; zero pad rows + slack for the panel / MFMA tile reads
        for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) lt[(size_t)j * d + i] = h->tlam[(size_t)i * d + j];
        if ((rc = dev_upload(h, &E.tgt.lamT, lt))) return rc;
    }
    if (h->tkind == TGT_EXPDATA || h->tkind == TGT_EXPCOLS) { if ((rc = dev_upload(h, &E.tgt.x, h->tx))) return rc; if ((rc = dev_upload(h,
        &E.tgt.y, h->ty))) return rc; }
    if (h->has_lo && (rc = dev_upload(h, &E.tgt.lo, h->tlo))) return rc;
    if (h->has_hi && (rc = dev_upload(h, &E.tgt.hi, h->thi))) return rc;
    if (h->has_pri) { if ((rc = dev_upload(h, &E.tgt.pmu, h->tpmu))) return rc; if ((rc = dev_upload(h, &E.tgt.psig, h->tpsig))) return rc;
        }
    if ((rc = dev_upload(h, &E.par0, h->par0))) return rc;
    if ((rc = dev_upload(h, &E.cmat0p, Cp))) return rc;

    // state
    const size_t L = (size_t)T * 64;
    if ((rc = dev_alloc(h, &E.theta, L * d))) return rc;
    // the user's functions on the host and at most sixteen tiles: candidate, results and flags in mapped host memory -- a phase kernel's
    // stores ARE the hand-over, the host's results are read by the next one in place (MCMCX_HOST_MAPPED=0: device buffers and copies)
    h->host_mapped = h->tkind == TGT_HOST && T <= 16 && h->sw.host_mapped != 0;
    // (RAM / SCAM use cs as sweep scratch)
    h->cs_mapped = h->host_mapped && (c.method == MCMCX_METHOD_DRAM || c.method == MCMCX_METHOD_ER);
    if ((rc = h->host_mapped ? host_alloc(h, &E.cand, L * d) : dev_alloc(h, &E.cand, L * d))) return rc;
    if ((rc = dev_alloc(h, &E.zs, L * 2 * d))) return rc;
    if ((rc = h->cs_mapped ? host_alloc(h, &E.cs, L * 2 * d) : dev_alloc(h, &E.cs, L * 2 * d))) return rc;
    E.xscr = nullptr;
    // step_kernel_pooled_dr_big's quadratic-form vectors; npar > 320: adapt_post_kernel's work vector
    if (((h->pooled && h->dodr) || d > 320) && (rc = dev_alloc(h, &E.xscr, L * 2 * d))) return rc;
    if ((rc = dev_alloc(h, &E.scal, L * NSCAL))) return rc;
    if ((rc = dev_alloc(h, &E.ictr, L * NICTR))) return rc;
    if ((rc = dev_alloc(h, &E.rngn, L))) return rc;
    E.R = nullptr;
    if (!h->pooled && (rc = dev_alloc(h, &E.R, L * P, false))) return rc;          // pooled: one shared factor instead
    if ((rc = dev_alloc(h, &E.basetheta, L * d))) return rc;
    if (h->usesvd && h->pooled && c.method == MCMCX_METHOD_SCAM) {     // pooled SCAM: one rotation for every chain
        // npar > 240: slower, not refused -- scam_kernel / scam_mw_kernel on per-chain copies
        h->scam_replicated = scam_pooled_lds(d) > 160 * 1024;
        if (h->scam_replicated) {
            if ((rc = dev_alloc(h, &E.Rf, L * (size_t)d * d, false))) return rc;
            if ((rc = dev_alloc(h, &E.qstd, L * d))) return rc;
        } else if ((rc = dev_alloc(h, &h->d_sharedU, 2 * shared_u_stride(h) + d, false))) return rc;
        h->pool_U = Rfull; h->pool_std = qstd0;
        if ((rc = upload_shared_u(h))) return rc;
    } else if (h->usesvd && h->pooled) {                // pooled AM with the SVD factor: one full matrix for every chain (below)
        h->pool_Rf = Rfull;
    } else if (h->usesvd) {
        const size_t DD = (size_t)d * d;
        if ((rc = dev_alloc(h, &E.Rf, L * DD, false))) return rc;
        if (c.method != MCMCX_METHOD_RAM) {                 // work space of the adaptation's SVD; RAM never refactors
            if ((rc = dev_alloc(h, &E.Gw, L * DD))) return rc;
            if ((rc = dev_alloc(h, &E.Vw, L * DD))) return rc;
            if (svd_blocked(h) && (c.doadapt != 0 || c.doburnin != 0)) {      // chain-major copies for svd_blocked_kernel
                if ((rc = dev_alloc(h, &h->d_Gc, L * DD, false))) return rc;
                if ((rc = dev_alloc(h, &h->d_Vc, L * DD, false))) return rc;
                if ((rc = dev_alloc(h, &h->d_svc, L * d, false))) return rc;
                if ((rc = dev_alloc(h, &h->d_need, L))) return rc;
                if ((rc = dev_alloc(h, &h->d_state, L))) return rc;
                if ((rc = dev_alloc(h, &h->d_anyrot, 1))) return rc;
            }
        }
        if ((rc = dev_alloc(h, &E.qstd, L * d))) return rc;
        if (h->dodr && (rc = dev_alloc(h, &E.R2f, L * DD, false))) return rc;
        if ((rc = dev_bcast(h, E.Rf, Rfull))) return rc;
        if ((rc = dev_bcast(h, E.qstd, qstd0))) return rc;
        if (h->dodr) {
            std::vector<double> r2 = Rfull;
            for (auto &v : r2) v = v / c.drscale;
            if ((rc = dev_bcast(h, E.R2f, r2))) return rc;
        }
    }
    E.R2 = E.iC = nullptr;
    if (h->dodr && !h->pooled) {                        // pooled: one R2 and one iC for every chain (below)
        if ((rc = dev_alloc(h, &E.R2, L * P, false))) return rc;
        if ((rc = dev_alloc(h, &E.iC, L * P, false))) return rc;
    }
    const bool am = (c.method != MCMCX_METHOD_RAM) && (c.doadapt != 0 || c.doburnin != 0) && !h->pooled;
    E.cmat = E.mean = E.Rtmp = nullptr; E.rowlist = nullptr;
    // history ring
    const bool need_hist = am || c.record_chain;
    h->wcap = 0; E.hist = E.s2hist = nullptr; E.wacc = nullptr; E.record_s2 = 0;
    if (need_hist) {
        // AP windows (adapthist > 1) may reach back to an arbitrarily old row: keep everything.  Otherwise the longest
        // window is the first AM one, rows 2 .. T1 with T1 the first multiple of adaptint / badaptint that is
        // >= burnintime + adaptint + adapthist (MCMC_adapt.F90:42-46,105): up to adaptint - 1 iterations past that
        // threshold when it is not a multiple itself (burn-in ticks only rescale and never restart the window)
        long long wc = (c.record_chain || (am && c.adapthist > 1)) ? (long long)c.nsimu + 1
                                      : (long long)c.burnintime + 2LL * c.adaptint + c.adapthist + 2;
        if (wc > (long long)c.nsimu + 1) wc = (long long)c.nsimu + 1;
        h->wcap = (int)wc;
        if ((rc = dev_alloc(h, &E.hist, L * (size_t)h->wcap * (d + ny), false))) return rc;
        if ((rc = dev_alloc(h, &E.wacc, (size_t)T * h->wcap))) return rc;
        if (c.record_chain && c.updatesigma) { E.record_s2 = 1; if ((rc = dev_alloc(h, &E.s2hist, L * (size_t)h->wcap * ny))) return rc; }
    }
    E.wcap = h->wcap > 0 ? h->wcap : 1;
    if (am) {
        if ((rc = dev_alloc(h, &E.cmat, L * P))) return rc;
        if ((rc = dev_alloc(h, &E.mean, L * d))) return rc;
        if ((rc = dev_alloc(h, &E.Rtmp, L * P))) return rc;
        if ((rc = dev_alloc(h, &E.rowlist, L * (size_t)(h->wcap + 1)))) return rc;
    }
    E.sharedR = nullptr;
    if (h->pooled) {
        if ((long long)c.nchains * (h->comm ? h->comm->nranks
            : 1) < 2) return fail(-8, "pooled mode needs at least 2 chains over all ranks");
        if (phase_cut(h) || (fused_cols(h) && c.method == MCMCX_METHOD_SCAM))
            return fail(-8, "pooled mode needs one of the single-launch device targets (gauss, banana, expdata, expcols; scam: not expcols)");
        if ((rc = dev_alloc(h, &h->d_sharedR, (size_t)P, false))) return rc;
        HIPCHK(hipMemcpy(h->d_sharedR, Rp.data(), (size_t)P * 8, hipMemcpyHostToDevice));
        E.sharedR = h->d_sharedR;
        h->pool_R = Rp; h->pool_C = Cp; h->pool_mean = h->par0; h->pool_W = (double)c.initcmatn;
        if (c.method == MCMCX_METHOD_DRAM || c.method == MCMCX_METHOD_RAM || c.method == MCMCX_METHOD_ER) {
            if ((rc = dev_alloc(h, &h->d_sharedRT, (size_t)((d + 3) & ~3) * d + PWS, false))) return rc;
            if ((rc = h->usesvd ? upload_shared_rf(h) : upload_shared_rt(h))) return rc;
            if (h->usesvd) E.sharedR = h->d_sharedRT;   // the lane-per-chain kernel reads the full matrix through the scalar cache
        }
    }
    E.hev = E.hx = nullptr;
    if (phased(h)) {
        if ((rc = h->host_mapped ? host_alloc(h, &E.hev, L * (NHE - 1 + ny)) : dev_alloc(h, &E.hev, L * (NHE - 1 + ny)))) return rc;
        if (ny > 1) {
            if ((rc = dev_alloc(h, &E.ssv, L * ny))) return rc;
            if ((rc = dev_alloc(h, &E.s2v, L * ny))) return rc;
            if ((rc = dev_alloc(h, &E.ss2v, L * ny))) return rc;
            if ((rc = dev_bcast(h, E.s2v, h->sigma2v))) return rc;
            std::vector<double> gs(ny);
            for (int j = 0; j < ny; ++j) gs[j] = c.N0 / 2.0 + (double)h->nobsv[j] / 2.0;
            if ((rc = dev_upload(h, &E.gshapev, gs))) return rc;
        }
        if ((rc = h->host_mapped ? host_alloc(h, &E.hx, L * NHX) : dev_alloc(h, &E.hx, L * NHX))) return rc;
    }
    E.accmask = nullptr;
    if (c.record_accept && (rc = dev_alloc(h, &E.accmask, (size_t)c.nsimu * T))) return rc;
    // the lane-group step kernel where it covers the configuration: MCMCX_GROUP = 1 / 0 forces it on / off (A/B, tests)
    h->group_d4 = 0;
    if (group_covers(h)) {
        const char *ev = getenv("MCMCX_GROUP");
        int ex = 0;
        const bool pow2 = h->dodr && c.drscale > 0.0 && std::frexp(c.drscale, &ex) == 0.5 && ex > -64 && ex < 64;
        // (the power-of-two form keeps iC in LDS: above npar 24 that leaves fewer waves per CU than the register form's four)
        const int drm = !h->dodr ? 0 : (pow2 && d <= 24 && !(getenv("MCMCX_GROUP_DR2") && atoi(getenv("MCMCX_GROUP_DR2")) == 0)) ? 2 : 1;
        const char *gwe = getenv("MCMCX_GROUP_GW");                                             // (4 / 16: A/B, tests)
        const int gw = (d <= 16 && gwe) ? (atoi(gwe) == 4 ? 4 : 16) : group_width(h);
        const bool on = ev ? atoi(ev) != 0 : group_wins(h, drm, gw);
        if (on) h->group_d4 = d <= 32 ? ((d + 3) & ~3) : ((d + 7) & ~7);
        if (h->group_d4 && (E.hist || E.accmask) && (rc = dev_alloc(h, &h->d_accb, L * (size_t)GROUP_MAXSEG))) return rc;
        if (h->group_d4) {
            h->group_gw = gw;
            h->group_drm = drm;
            h->group_check_due = true;
            if (h->group_drm == 2 && (rc = dev_alloc(h, &h->d_gflag, 1))) return rc;
        }
    }
    // method = 'ram' with few chains on the lane-group kernel (MCMCX_RAM_GROUP = 1 / 0: always where it covers / never; A/B, tests)
    h->ram_group_d4 = 0;
    if (ram_group_covers(h)) {
        const char *rg = getenv("MCMCX_RAM_GROUP");
        if (!rg) rg = getenv("MCMCX_GROUP");               // (the kernel-family switch of the tests covers it too)
        if (rg ? atoi(rg) != 0 : ram_group_wins(h)) {
            h->ram_group_d4 = d <= 16 ? 16 : d <= 32 ? 32 : d <= 56 ? 56 : 64;
            if ((E.hist || E.accmask) && (rc = dev_alloc(h, &h->d_accb, L * (size_t)GROUP_MAXSEG))) return rc;
        }
    }
    // the adaptation's factorisation (Cholesky branch) with the matrices in LDS (tile_factor_kernel): any chain count, npar <= 32 -- config
    // 3's size (npar 20, delayed rejection, 262144 chains): 2.11 -> 1.44 ms per tick, bound by VALU issue (~25 instructions per inner step
    // of four chains). Above npar 32 adapt_post_kernel's 8 x 8 register blocks stay: at npar 50 x 1 048 576 chains they stream the matrices
    // ~5.6 times (16.0 ms) and still beat the LDS form, whose three waves per CU issue ~15 x the instructions per chain (35.2 ms;
    // profiles/r05_a/tick_ab.txt). MCMCX_TILE_FACTOR = 0 / 1: never / up to npar 64 (test switch: both forms on one problem).
    {
        const char *tf = getenv("MCMCX_TILE_FACTOR");
        h->tile_factor = am && !h->usesvd && d <= ((tf && atoi(tf) == 1) ? 64 : 32) && !(tf && atoi(tf) == 0);
    }
    // 1/simuind**nuparam, computed like the reference: real(simuind) is default REAL (MCMC_run_ram.F90:166)
    {
        std::vector<double> rs((size_t)c.nsimu + 2, 0.0);
        for (int it = 1; it <= c.nsimu; ++it) rs[it] = 1.0 / std::pow((double)(float)it, c.nuparam);
        const double *p = nullptr;
        if ((rc = dev_upload(h, &p, rs))) return rc;
        h->d_ramscale = const_cast<double *>(p);
    }
    // longest pooled vector: 2 + d + P
    if ((rc = dev_alloc(h, &h->d_moments, (size_t)(T + 1) * (2 + d + P)))) return rc;
    // longest vector + the stop flag
    if ((rc = dev_alloc(h, &h->d_gather, (size_t)(h->comm ? h->comm->nranks : 1) * (3 + d + P)))) return rc;
    if ((rc = dev_alloc(h, &h->d_pooled, (size_t)(3 + d + P)))) return rc;

    // fill theta = par0, R = R(cmat0), chaincmat = cmat0, chainmean = par0, scalars
    {
        std::vector<double> sc(L * NSCAL, 0.0);
        std::vector<uint32_t> ic(L * NICTR, 0u);
        for (int t = 0; t < T; ++t)
            for (int l = 0; l < 64; ++l) {
                sc[((size_t)t * NSCAL + S_SIGMA2) * 64 + l] = h->sigma2;
                sc[((size_t)t * NSCAL + S_WSUM) * 64 + l] = (double)c.initcmatn;
                ic[((size_t)t * NICTR + I_CHAININD) * 64 + l] = 1;
                ic[((size_t)t * NICTR + I_CURCOUNT) * 64 + l] = 1;
                ic[((size_t)t * NICTR + I_BASECNT) * 64 + l] = 1;
                ic[((size_t)t * NICTR + I_WINSTART) * 64 + l] = 2;
            }
        HIPCHK(hipMemcpyAsync(E.scal, sc.data(), sc.size() * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(E.ictr, ic.data(), ic.size() * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if ((rc = dev_bcast(h, E.theta, h->par0))) return rc;
        if (!h->pooled && (rc = dev_bcast(h, E.R, Rp))) return rc;
        if (h->dodr) {                                   // iC = dpotri(R), R2 = R/drscale, MCMC_adapt.F90:216-225
            std::vector<double> iCp = Rp, R2p(P);
            if (h->usesvd) for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i) iCp[h_pidx(i, j, d)] = Rfull[(size_t)j * d + i];
            if (host_potri(d, iCp) != 0) return fail(-34, "ERROR: cannot invert cmat");
            for (int e = 0; e < P; ++e) R2p[e] = Rp[e] / c.drscale;
            if (h->pooled) {
                h->pool_iC = iCp;
                if (h->usesvd) { h->pool_R2 = Rfull; for (auto &v : h->pool_R2) v = v / c.drscale; } else h->pool_R2 = R2p;
                if ((rc = dev_alloc(h, &h->d_sharedR2, h->usesvd ? (size_t)((d + 3) & ~3) * d + PWS : (size_t)P, false))) return rc;
                if ((rc = dev_alloc(h, &h->d_sharediC, (size_t)P, false))) return rc;
                if (c.method == MCMCX_METHOD_DRAM && h->d_sharedRT && pooled_mfma_lds(d) <= 160 * 1024 &&
                    !(h->sw.pooled_scalar > 0)) {          // the second stage on the matrix cores too
                    if ((rc = dev_alloc(h, &h->d_sharedR2T, (size_t)((d + 3) & ~3) * d + PWS, false))) return rc;
                    if ((rc = dev_alloc(h, &h->d_sharediCd, (size_t)((d + 3) & ~3) * d + PWS, false))) return rc;
                }
                if ((rc = pooled_upload_dr(h, false))) return rc;
            } else {
            if ((rc = dev_bcast(h, E.R2, R2p))) return rc;
            if ((rc = dev_bcast(h, E.iC, iCp))) return rc;
            }
        }
        if (am) {
            if ((rc = dev_bcast(h, E.cmat, Cp))) return rc;
            if ((rc = dev_bcast(h, E.mean, h->par0))) return rc;
        }
    }
    if (h->external) {                                  // MCMC_run1: every evaluation is the caller's, the first one included
        if (h->pooled) return fail(-8, "mcmcx_set_target_external: not in pooled mode");
        if ((rc = dev_alloc(h, &h->d_r1, L * (size_t)(3 * d + 3 * ny + NR1)))) return rc;
    } else if (phased(h)) {                             // first point: sspri1, ss1 from the callbacks (MCMC_run.F90:35-36)
        int rc2 = host_eval(h, E.theta, d, false);
        if (rc2) return rc2;
    }
    launch_init(h);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    h->simuind = 1;
    h->inited = true;
    return 0;
}

int mcmcx_install_signal_handlers(void)
{
    struct sigaction act;
    memset(&act, 0, sizeof(act));
    act.sa_handler = on_signal;
    sigemptyset(&act.sa_mask);
    const int sigs[] = {SIGHUP, SIGINT, SIGTERM, SIGTSTP, SIGUSR1, SIGUSR2};
    for (int sg : sigs) if (sigaction(sg, &act, nullptr) != 0) return fail(-1, "sigaction failed");
    g_sig_installed = true;
    return 0;
}
int mcmcx_interrupted(void) { return g_interrupt ? 1 : 0; }
void mcmcx_clear_interrupt(void) { g_interrupt = 0; }

static int run_impl(mcmcx_handle h, int32_t upto);
int mcmcx_run(mcmcx_handle h, int32_t upto)
{
    const int rc = run_impl(h, upto);
    // Several ranks meet in this engine's ticks: one that fails must not leave the others waiting in the next gather --
    // the communicator is marked failed, which the peers' waits poll (comm_wait_stream, shm_barrier)
    // -- once the run has entered its loop, that is: an argument or state error returned before that (-40 not inited, -41 external
    // target) strands nobody, and marking the communicator for it would abort peers that have nothing to do with it
    if (rc < 0 && h && h->inited && h->run_entered && collective_run(h)) comm_mark_failed(h->comm);
    if (h) h->run_entered = false;
    return rc;
}
static int run_impl(mcmcx_handle h, int32_t upto)
{
    if (!h) return fail(-1, "null handle");
    if (!h->inited) return fail(-40, "we have not inited");                           // MCMC_run.F90:22
    if (h->external) return fail(-41, "mcmcx_run: the target is external (mcmcx_set_target_external): drive the chain with mcmcx_run1_*");
    // (with the phases fused, iteration it + 1's proposal -- Philox draws included -- has already run when an error surfaces in iteration
    // it's evaluation; a second run on the handle would draw that proposal again and leave the reference's stream order silently: ADVICE
    // round 5)
    if (h->failed) return fail(-42, "mcmcx_run: an earlier run on this handle failed inside a host-callback iteration; its chains are in an undefined state -- destroy the handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    const mcmcx_config &c = h->cfg;
    if (upto > c.nsimu) upto = c.nsimu;
    const int maxseg = (h->group_d4 || h->ram_group_d4) ? GROUP_MAXSEG : (c.method == MCMCX_METHOD_RAM && !h->pooled) ? 4096 : 1 << 30;
    int it = h->simuind + 1;
    // Several ranks that meet in this engine's ticks (pooled mode with a communicator): a rank that left the loop alone --
    // on a signal it happened to see first, or on an error of its own -- would leave its peers waiting in the next gather.
    // So a signal only raises this rank's stop flag in the exchanged vector and the run is left at the tick where every
    // rank reads the summed flag (pooled_reduce); an error marks the communicator failed (mcmcx_run).
    const bool coll = collective_run(h);
    h->run_entered = true;
    // ... which needs a tick ahead.  Whether one lies in (it, upto] follows from the configuration alone, the same on every
    // rank: past the last one (doadapt = 0, iterations beyond adaptend, the tail of a run) no collective is left to strand a
    // peer in, and a signal is acted on at the next launch boundary like in a run of one rank.
    int last_tick = 0;
    if (coll) for (int i2 = upto; i2 >= it; --i2) if (adapt_mode(c, i2) != 0 || pooled_ram_due(h, i2)) { last_tick = i2; break; }
    h->stop_seen = false;
    while (it <= upto) {
        if (g_interrupt && (!coll || it > last_tick)) {     // a caught signal: stop at this launch boundary
            int rc = mcmcx_sync(h); if (rc) return rc;
            h->simuind = it - 1;
            return MCMCX_INTERRUPTED;
        }
        int end = it, mode = 0;
        bool ramtick = false;
        for (;; ++end) {                                    // extend the launch up to the next tick
            mode = adapt_mode(c, end);
            ramtick = pooled_ram_due(h, end);
            if (mode != 0 || ramtick || end == upto || end - it + 1 >= maxseg) break;
        }
        if (phase_cut(h)) {
            for (int i2 = it; i2 <= end; ++i2) { int rc = host_iteration(h, i2, i2 < end); if (rc) { h->failed = true; return rc; } }
        } else {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            hipError_t er = hipEventCreate(&e0);
            if (er == hipSuccess) er = hipEventCreate(&e1);
            if (er == hipSuccess) er = hipEventRecord(e0, h->stream);
            if (er == hipSuccess) {
                h->launch_err.clear();
                if (h->cfg.method == MCMCX_METHOD_SCAM) launch_scam(h, it, end); else launch_step(h, it, end);
                er = hipGetLastError();
            }
            // a cover predicate and the dispatch disagree: fail, never skip iterations silently
            if (er == hipSuccess && !h->launch_err.empty()) {
                if (e0) (void)hipEventDestroy(e0);
                if (e1) (void)hipEventDestroy(e1);
                return fail(-103, "step launch: " + h->launch_err);
            }
            if (er == hipSuccess) er = hipEventRecord(e1, h->stream);
            if (er != hipSuccess) {                       // nothing is left behind on the error path
                if (e0) (void)hipEventDestroy(e0);
                if (e1) (void)hipEventDestroy(e1);
                return fail(-100, std::string("step launch: ") + hipGetErrorString(er));
            }
            h->pending.emplace_back(e0, e1);
            h->launches += 1; h->steps += (end - it + 1);
        }
        int trc = 0;
        if (mode != 0) {
            if (h->pooled) trc = pooled_tick(h, end, mode);
            else { launch_adapt(h, end, mode); h->group_check_due = true; HIPCHK(hipGetLastError()); }
        }
        if (ramtick && trc == 0) trc = pooled_ram_tick(h, end);
        if (trc) return trc;
        if (h->stop_seen) {                                 // every rank read the same summed stop flag at this tick: leave together,
            h->stop_seen = false;                           // after iteration `end` WITH its adaptation applied (a resumed run
            int rc = mcmcx_sync(h); if (rc) return rc;      // continues at end + 1 like an uninterrupted one)
            h->simuind = end;
            return MCMCX_INTERRUPTED;
        }
        it = end + 1;
        if (g_sig_installed) { HIPCHK(hipStreamSynchronize(h->stream)); h->simuind = std::max(h->simuind, end); }
        if (h->pending.size() > 4096) { int rc = mcmcx_sync(h); if (rc) return rc; }
    }
    h->simuind = std::max(h->simuind, (int)upto);
    return 0;
}

int mcmcx_run1_decide(mcmcx_handle h, int32_t drstage, const double *oldpar2, const double *ssprev2, const double *sspri2,
                      const double *oldpar1, const double *ssprev1, const double *sspri1, const double *alpha12,
                      const double *newpar, const double *ss, const double *sspri, double *alpha_out, int32_t *reject_out)
{
    int rc = run1_check(h); if (rc) return rc;
    if (!oldpar1 || !ssprev1 || !sspri1 || !newpar || !ss || !sspri || !alpha_out
        || !reject_out) return fail(-1, "mcmcx_run1_decide: null argument");
    const bool dr2 = drstage > 1 && h->dodr;
    if (dr2 && (!oldpar2 || !ssprev2 || !sspri2
        || !alpha12)) return fail(-1, "mcmcx_run1_decide: the second stage needs oldpar2, ssprev2, sspri2, alpha12");
    const int d = h->d, ny = h->ny, n = h->cfg.nchains, s0 = 3 * d + 3 * ny;
    h->h_r1.assign((size_t)h->ntiles * 64 * (s0 + NR1), 0.0);
    run1_put(h, 0, d, dr2 ? oldpar2 : nullptr); run1_put(h, d, d, oldpar1); run1_put(h, 2 * d, d, newpar);
    run1_put(h, 3 * d, ny, dr2 ? ssprev2 : nullptr); run1_put(h, 3 * d + ny, ny, ssprev1); run1_put(h, 3 * d + 2 * ny, ny, ss);
    run1_put(h, s0 + R1_PRI2, 1, dr2 ? sspri2 : nullptr); run1_put(h, s0 + R1_PRI1, 1, sspri1); run1_put(h, s0 + R1_PRI, 1, sspri);
    run1_put(h, s0 + R1_A12, 1, dr2 ? alpha12 : nullptr);
    if ((rc = run1_launch<0>(h, drstage))) return rc;
    run1_get(h, s0 + R1_ALPHA, 1, alpha_out);
    std::vector<double> rj(n);
    run1_get(h, s0 + R1_REJECT, 1, rj.data());
    for (int c = 0; c < n; ++c) reject_out[c] = rj[c] != 0.0 ? 1 : 0;
    return 0;
}

int mcmcx_run1_propose(mcmcx_handle h, int32_t stage, const double *from, double *newpar_out)
{
    int rc = run1_check(h); if (rc) return rc;
    if (!from || !newpar_out) return fail(-1, "mcmcx_run1_propose: null argument");
    const int d = h->d, ny = h->ny;
    h->h_r1.assign((size_t)h->ntiles * 64 * (3 * d + 3 * ny + NR1), 0.0);
    run1_put(h, 0, d, from);
    if (stage > 1 && h->dodr) rc = run1_launch<2>(h, 2); else rc = run1_launch<1>(h, 1);
    if (rc) return rc;
    run1_get(h, 2 * d, d, newpar_out);
    return 0;
}

int mcmcx_run1_sscrit(mcmcx_handle h, const double *ssprev1, const double *sspri1, double *sscrit_out)
{
    int rc = run1_check(h); if (rc) return rc;
    if (!ssprev1 || !sspri1 || !sscrit_out) return fail(-1, "mcmcx_run1_sscrit: null argument");
    const int d = h->d, ny = h->ny, s0 = 3 * d + 3 * ny;
    h->h_r1.assign((size_t)h->ntiles * 64 * (s0 + NR1), 0.0);
    run1_put(h, 3 * d + ny, ny, ssprev1); run1_put(h, s0 + R1_PRI1, 1, sspri1);
    if ((rc = run1_launch<3>(h, 1))) return rc;
    run1_get(h, s0 + R1_CRIT, 1, sscrit_out);
    return 0;
}

int mcmcx_sync(mcmcx_handle h)
{
    if (!h) return fail(-1, "null handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (auto &p : h->pending) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, p.first, p.second));
        h->ms_total += ms;
        (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second);
    }
    h->pending.clear();
    return 0;
}

int mcmcx_kernel_time(mcmcx_handle h, double *ms, int64_t *launches, int64_t *steps, int reset)
{
    int rc = mcmcx_sync(h);
    if (rc) return rc;
    if (ms) *ms = h->ms_total;
    if (launches) *launches = h->launches;
    if (steps) *steps = h->steps;
    if (reset) { h->ms_total = 0.0; h->launches = 0; h->steps = 0; }
    return 0;
}

int32_t mcmcx_simuind(mcmcx_handle h) { return h ? h->simuind : -1; }



int mcmcx_get_counters(mcmcx_handle h, int32_t chain, int32_t *out)
{
    int rc = check_chain(h, chain); if (rc) return rc;
    std::vector<uint32_t> v;
    if ((rc = fetch_chain_vec(h, h->E.ictr, NICTR, chain, v))) return rc;
    out[0] = (int32_t)v[I_STAYED]; out[1] = (int32_t)v[I_BNDSTAYED]; out[2] = (int32_t)v[I_DRACC]; out[3] = (int32_t)v[I_DRTRIES];
    out[4] = (int32_t)v[I_CHAININD]; out[5] = (int32_t)v[I_STATUS]; out[6] = (int32_t)v[I_ERSTAYED]; out[7] = (int32_t)v[I_CURCOUNT];
    return 0;
}

int mcmcx_get_totals_n(mcmcx_handle h, int64_t *out, int32_t n)
{
    if (!out || n < 1) return fail(-1, "mcmcx_get_totals_n: bad argument");
    int64_t t7[7];
    int rc = mcmcx_get_totals(h, t7); if (rc) return rc;
    for (int i = 0; i < n; ++i) out[i] = i < 7 ? t7[i] : 0;
    return 0;
}

int mcmcx_get_totals(mcmcx_handle h, int64_t *t7)
{
    if (!h || !h->inited) return fail(-40, "we have not inited");
    std::vector<uint32_t> v;
    int rc = fetch(h, h->E.ictr, (size_t)h->nlanes * NICTR, v); if (rc) return rc;
    for (int i = 0; i < 7; ++i) t7[i] = 0;
    for (int c = 0; c < h->cfg.nchains; ++c) {
        int t = c / 64, l = c % 64;
        auto at = [&](int k) { return (int64_t)v[((size_t)t * NICTR + k) * 64 + l]; };
        t7[0] += at(I_STAYED); t7[1] += at(I_BNDSTAYED); t7[2] += at(I_DRACC); t7[3] += at(I_DRTRIES);
        t7[5] += at(I_DOWNS); t7[6] |= at(I_STATUS);
    }
    t7[6] |= h->pool_status;                            // pooled RAM: a tick whose Gram matrix was not positive definite was skipped
    // proposals evaluated: one per iteration (d componentwise ones with method='scam') + the delayed-rejection tries
    t7[4] = (int64_t)h->cfg.nchains * (int64_t)(h->simuind - 1) * (h->cfg.method == MCMCX_METHOD_SCAM ? h->d : 1) + t7[3];
    return 0;
}

int mcmcx_get_theta(mcmcx_handle h, double *out)
{
    if (!h || !h->inited) return fail(-40, "we have not inited");
    std::vector<double> v;
    int rc = fetch(h, h->E.theta, (size_t)h->nlanes * h->d, v); if (rc) return rc;
    for (int c = 0; c < h->cfg.nchains; ++c)
        for (int k = 0; k < h->d; ++k) out[(size_t)c * h->d + k] = v[((size_t)(c / 64) * h->d + k) * 64 + (c % 64)];
    return 0;
}

int mcmcx_get_scalars(mcmcx_handle h, double *out)
{
    if (!h || !h->inited) return fail(-40, "we have not inited");
    std::vector<double> v;
    int rc = fetch(h, h->E.scal, (size_t)h->nlanes * NSCAL, v); if (rc) return rc;
    const int idx[4] = {S_SS1, S_PRI1, S_SIGMA2, S_ALPHA12};
    for (int c = 0; c < h->cfg.nchains; ++c)
        for (int k = 0; k < 4; ++k) out[(size_t)c * 4 + k] = v[((size_t)(c / 64) * NSCAL + idx[k]) * 64 + (c % 64)];
    return 0;
}

int mcmcx_get_rng(mcmcx_handle h, int32_t chain, uint64_t *n, int32_t *saved, double *saved_y)
{
    int rc = check_chain(h, chain); if (rc) return rc;
    std::vector<uint64_t> vn; std::vector<uint32_t> vi; std::vector<double> vs;
    if ((rc = fetch_chain_vec(h, h->E.rngn, 1, chain, vn))) return rc;
    if ((rc = fetch_chain_vec(h, h->E.ictr, NICTR, chain, vi))) return rc;
    if ((rc = fetch_chain_vec(h, h->E.scal, NSCAL, chain, vs))) return rc;
    if (n) *n = vn[0];
    if (saved) *saved = (int32_t)vi[I_SAVED];
    if (saved_y) *saved_y = vs[S_SAVEDY];
    return 0;
}

int mcmcx_get_R(mcmcx_handle h, int32_t chain, double *R)
{
    int rc = check_chain(h, chain); if (rc) return rc;
    std::vector<double> p;
    if (h->pooled && h->usesvd) { const auto &M = h->cfg.method == MCMCX_METHOD_SCAM ? h->pool_U : h->pool_Rf; memcpy(R, M.data(),
        sizeof(double) * M.size()); return 0; }
    if (h->pooled) { unpack_upper(h->d, h->pool_R, R, false); return 0; }
    if (h->usesvd) {                                     // full column-major factor
        if ((rc = fetch_chain_vec(h, h->E.Rf, h->d * h->d, chain, p))) return rc;
        memcpy(R, p.data(), sizeof(double) * p.size());
        return 0;
    }
    if ((rc = fetch_chain_vec(h, h->E.R, h->P, chain, p))) return rc;
    unpack_upper(h->d, p, R, false);
    return 0;
}

int mcmcx_get_qcovstd(mcmcx_handle h, int32_t chain, double *std)
{
    int rc = check_chain(h, chain); if (rc) return rc;
    if (h->pooled && h->usesvd && h->cfg.method == MCMCX_METHOD_SCAM) { memcpy(std, h->pool_std.data(),
        sizeof(double) * h->pool_std.size()); return 0; }
    if (!h->E.qstd) return fail(-45, "no SVD state (condmax = 0)");
    std::vector<double> p;
    if ((rc = fetch_chain_vec(h, h->E.qstd, h->d, chain, p))) return rc;
    memcpy(std, p.data(), sizeof(double) * p.size());
    return 0;
}

int mcmcx_get_dr(mcmcx_handle h, int32_t chain, double *R2, double *iC)
{
    int rc = check_chain(h, chain); if (rc) return rc;
    if (!h->dodr) return fail(-43, "drscale = 0: no delayed-rejection state");
    std::vector<double> p;
    if (h->pooled) {                                    // one pair of tables for every chain
        if (R2 && h->usesvd) memcpy(R2, h->pool_R2.data(), sizeof(double) * h->pool_R2.size());
        else if (R2) unpack_upper(h->d, h->pool_R2, R2, false);
        if (iC) unpack_upper(h->d, h->pool_iC, iC, false);
        return 0;
    }
    if (R2 && h->usesvd) { if ((rc = fetch_chain_vec(h, h->E.R2f, h->d * h->d, chain, p))) return rc; memcpy(R2, p.data(),
        sizeof(double) * p.size()); }
    else if (R2) { if ((rc = fetch_chain_vec(h, h->E.R2, h->P, chain, p))) return rc; unpack_upper(h->d, p, R2, false); }
    if (iC) { if ((rc = fetch_chain_vec(h, h->E.iC, h->P, chain, p))) return rc; unpack_upper(h->d, p, iC, false); }
    return 0;
}

int mcmcx_get_chaincov(mcmcx_handle h, int32_t chain, double *cmat, double *mean, double *wsum)
{
    int rc = check_chain(h, chain); if (rc) return rc;
    std::vector<double> s;
    if ((rc = fetch_chain_vec(h, h->E.scal, NSCAL, chain, s))) return rc;
    if (wsum) *wsum = s[S_WSUM];
    if (!h->E.cmat) {                                   // no AM state on the device: chaincmat = cmat0, chainmean = par0
        if (cmat) memcpy(cmat, h->cmat0.data(), sizeof(double) * (size_t)h->d * h->d);
        if (mean) memcpy(mean, h->par0.data(), sizeof(double) * (size_t)h->d);
        return 0;
    }
    std::vector<double> p, m;
    if ((rc = fetch_chain_vec(h, h->E.cmat, h->P, chain, p))) return rc;
    if ((rc = fetch_chain_vec(h, h->E.mean, h->d, chain, m))) return rc;
    if (cmat) unpack_upper(h->d, p, cmat, true);
    if (mean) memcpy(mean, m.data(), sizeof(double) * (size_t)h->d);
    return 0;
}

int mcmcx_get_accept_masks(mcmcx_handle h, uint64_t *masks, int32_t *ntiles)
{
    if (!h || !h->inited) return fail(-40, "we have not inited");
    if (ntiles) *ntiles = h->ntiles;
    if (!h->E.accmask) return fail(-42, "record_accept was not requested");
    if (!masks) return 0;
    std::vector<uint64_t> v;
    int rc = fetch(h, h->E.accmask, (size_t)h->simuind * h->ntiles, v); if (rc) return rc;
    memcpy(masks, v.data(), v.size() * 8);
    return 0;
}

int mcmcx_get_accepted(mcmcx_handle h, int32_t chain, uint8_t *acc)
{
    int rc = check_chain(h, chain); if (rc) return rc;
    const int tile = chain / 64, lane = chain % 64;
    if (h->E.accmask) {
        std::vector<uint64_t> v;
        if ((rc = fetch(h, h->E.accmask, (size_t)h->simuind * h->ntiles, v))) return rc;
        for (int it = 1; it <= h->simuind; ++it) acc[it - 1] = (uint8_t)((v[(size_t)(it - 1) * h->ntiles + tile] >> lane) & 1ull);
        return 0;
    }
    if (h->cfg.record_chain && h->E.wacc) {
        std::vector<uint64_t> v;
        if ((rc = fetch(h, h->E.wacc + (size_t)tile * h->wcap, (size_t)h->wcap, v))) return rc;
        for (int it = 1; it <= h->simuind; ++it) acc[it - 1] = (uint8_t)((v[it % h->wcap] >> lane) & 1ull);
        return 0;
    }
    return fail(-42, "record_accept / record_chain was not requested");
}

int mcmcx_get_chain(mcmcx_handle h, int32_t chain, double *chain_out, double *ss_out, double *s2_out, int32_t *nrows)
{
    int rc = check_chain(h, chain); if (rc) return rc;
    if (!h->cfg.record_chain || !h->E.hist) return fail(-42, "record_chain was not requested");
    const int tile = chain / 64, lane = chain % 64, d = h->d, W = h->wcap;
    std::vector<uint64_t> m;
    if ((rc = fetch(h, h->E.wacc + (size_t)tile * W, (size_t)W, m))) return rc;
    // this chain's lane of the history ring, gathered on the device: hv[slot*(d+1) + k]
    auto gather = [&](const double *src, size_t n, std::vector<double> &dst) -> int {
        double *tmp = nullptr;
        HIPCHK(hipMalloc(&tmp, n * sizeof(double)));       // (freed on every path below)
        hipLaunchKernelGGL(gather_lane_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, h->stream, src, tmp,
            n, lane);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        dst.resize(n);
        if (e == hipSuccess) e = hipMemcpy(dst.data(), tmp, n * sizeof(double), hipMemcpyDeviceToHost);
        (void)hipFree(tmp);
        if (e != hipSuccess) return fail(-100, hipGetErrorString(e));
        return 0;
    };
    std::vector<double> hv;
    const int ny = h->ny, hs = d + ny;                  // history row: theta, ss per column
    if ((rc = gather(h->E.hist + (size_t)tile * W * hs * 64, (size_t)W * hs, hv))) return rc;
    int row = -1;
    for (int it = 1; it <= h->simuind; ++it) {
        const size_t so = (size_t)(it % W) * hs;
        if ((m[it % W] >> lane) & 1ull) {
            ++row;
            if (chain_out) {
                for (int k = 0; k < d; ++k) chain_out[(size_t)row * (d + 1) + k] = hv[so + k];
                chain_out[(size_t)row * (d + 1) + d] = 1.0;
            }
            if (ss_out) { for (int j = 0; j < ny; ++j) ss_out[(size_t)row * (ny + 1) + j] = hv[so + d + j];
                ss_out[(size_t)row * (ny + 1) + ny] = 1.0; }
        } else {
            if (chain_out) chain_out[(size_t)row * (d + 1) + d] += 1.0;
            if (ss_out) ss_out[(size_t)row * (ny + 1) + ny] += 1.0;
        }
    }
    if (nrows) *nrows = row + 1;
    if (s2_out && h->E.s2hist) {
        std::vector<double> sv;
        if ((rc = gather(h->E.s2hist + (size_t)tile * W * ny * 64, (size_t)W * ny, sv))) return rc;
        for (int it = 1; it <= h->simuind; ++it) for (int j = 0; j < ny;
            ++j) s2_out[(size_t)(it - 1) * ny + j] = sv[(size_t)(it % W) * ny + j];
    }
    return 0;
}

int32_t mcmcx_pooled_moments_len(mcmcx_handle h) { return h ? 1 + h->d + h->P : -1; }

static int pooled_vec_len(const mcmcx_engine *h, int kind) { return kind == 2 ? 2 + h->P : 1 + h->d + h->P + (kind == 1 ? 1 : 0); }

static int pooled_moments_launch(mcmcx_engine *h, double *dev_dst, int kind, int it)
{   // (declared above)
    if (!h || !h->inited) return fail(-40, "we have not inited");
    HIPCHK(hipSetDevice(h->cfg.device));
    const int len = pooled_vec_len(h, kind), T = h->ntiles;
    const double rs = (kind == 2) ? 1.0 / std::pow((double)(float)it, h->cfg.nuparam) : 0.0;      // like d_ramscale (MCMC_run_ram.F90:166)
    const size_t mlds = ((size_t)64 * (h->d | 1) + 320) * sizeof(double);
    if (mlds <= 160 * 1024) hipLaunchKernelGGL(moments_kernel<false>, dim3(T), dim3(256), mlds, h->stream, h->E, h->d_moments,
        h->cfg.nchains, kind, it, rs);
    // npar >= 316: (64 (d | 1) + 320) 8 bytes exceed 160 KiB (317: 164 864)
    else hipLaunchKernelGGL(moments_kernel<true>, dim3(T), dim3(256), (size_t)320 * sizeof(double), h->stream, h->E, h->d_moments,
        h->cfg.nchains, kind, it, rs);
    for (long long stride = 1;; stride *= 64) {                            // six levels of the fixed pairwise tree per launch
        const long long groups = (T + 64 * stride - 1) / (64 * stride);
        hipLaunchKernelGGL(moments_tree_kernel, dim3((len + 255) / 256, (unsigned)groups), dim3(256), 0, h->stream, h->d_moments, T, len,
                           (int)stride, dev_dst);
        if (groups == 1) break;
    }
    HIPCHK(hipGetLastError());
    return 0;
}

int mcmcx_pooled_moments(mcmcx_handle h, double *out)
{
    int rc = pooled_moments_launch(h, nullptr, 0, 0); if (rc) return rc;
    std::vector<double> v;
    if ((rc = fetch(h, h->d_moments, (size_t)(1 + h->d + h->P), v))) return rc;
    memcpy(out, v.data(), sizeof(double) * v.size());
    return 0;
}

int mcmcx_set_exchange(mcmcx_handle h, mcmcx_exchange_t fn, void *user, void *dev_buf)
{
    if (!h) return fail(-1, "null handle");
    if (fn && !dev_buf) return fail(-1, "mcmcx_set_exchange: dev_buf is required");
    h->xfn = fn; h->xuser = user; h->xbuf = (double *)dev_buf;
    return 0;
}

int mcmcx_get_pooled(mcmcx_handle h, double *cmat, double *mean, double *wsum, double *R)
{
    if (!h || !h->inited) return fail(-40, "we have not inited");
    if (!h->pooled) return fail(-44, "not in pooled mode");
    if (cmat) unpack_upper(h->d, h->pool_C, cmat, true);
    if (mean) memcpy(mean, h->pool_mean.data(), sizeof(double) * (size_t)h->d);
    if (wsum) *wsum = h->pool_W;
    // scam: the rotation U; condmax > 0: the full SVD factor; column-major
    if (R && h->usesvd) { const auto &M = h->cfg.method == MCMCX_METHOD_SCAM ? h->pool_U : h->pool_Rf; memcpy(R, M.data(),
        sizeof(double) * M.size()); }
    else if (R) unpack_upper(h->d, h->pool_R, R, false);
    return 0;
}

/* same, result left in device memory at dev_out (e.g. the buffer an RCCL all-reduce works on);
 * asynchronous on the engine's stream: call mcmcx_sync before another stream reads it */
int mcmcx_pooled_moments_dev(mcmcx_handle h, void *dev_out)
{
    if (!dev_out) return fail(-1, "null argument");
    return pooled_moments_launch(h, (double *)dev_out, 0, 0);
}

// ------------------------------------------------------------------ the node: several GPUs, one communicator
int mcmcx_set_comm(mcmcx_handle h, mcmcx_comm_t c)
{
    if (!h) return fail(-1, "null handle");
    if (h->inited) return fail(-1, "mcmcx_set_comm after mcmcx_init");
    if (c && c->device != h->cfg.device) return fail(-1, "mcmcx_set_comm: the communicator lives on device " + std::to_string(c->device)
        + ", the engine on " + std::to_string(h->cfg.device));
    h->comm = c;
    return 0;
}

int mcmcx_allreduce_moments(mcmcx_handle h, double *host_out)
{
    if (!h || !h->inited) return fail(-40, "we have not inited");
    if (h->comm && h->comm->single_process && h->comm->nranks > 1)
        return fail(-1, "mcmcx_allreduce_moments: this communicator drives several GPUs from one process; use mcmcx_allreduce_moments_all");
    int rc = allreduce_moments_enqueue(h, 0); if (rc) return rc;
    if (host_out) {
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(host_out, h->d_pooled, (size_t)(1 + h->d + h->P) * 8, hipMemcpyDeviceToHost));
    }
    return 0;
}

int mcmcx_allreduce_moments_all(mcmcx_handle *hs, int32_t n, double *host_out)
{
    if (!hs || n < 1) return fail(-1, "bad argument");
    for (int i = 0; i < n; ++i) if (!hs[i] || !hs[i]->inited) return fail(-40, "we have not inited");
    int rc;
    for (int i = 0; i < n; ++i) if ((rc = allreduce_moments_enqueue(hs[i], 1))) return rc;
    NCCLCHK(ncclGroupStart());                                  // one thread, several devices: the gathers go out as a group
    for (int i = 0; i < n; ++i) if ((rc = allreduce_moments_enqueue(hs[i], 2))) { (void)ncclGroupEnd(); return rc; }
    NCCLCHK(ncclGroupEnd());
    for (int i = 0; i < n; ++i) if ((rc = allreduce_moments_enqueue(hs[i], 3))) return rc;
    if (host_out) {
        HIPCHK(hipSetDevice(hs[0]->cfg.device));
        HIPCHK(hipStreamSynchronize(hs[0]->stream));
        HIPCHK(hipMemcpy(host_out, hs[0]->d_pooled, (size_t)(1 + hs[0]->d + hs[0]->P) * 8, hipMemcpyDeviceToHost));
    }
    return 0;
}

// mcmcx_run on n engines at once, one host thread per engine (each GPU has its own stream; in pooled mode the threads
// meet in the all-gather of every adaptation tick).  User host callbacks are not thread-safe: engines with
// a host target run one after the other on the calling thread instead.
int mcmcx_run_all(mcmcx_handle *hs, int32_t n, int32_t upto)
{
    if (!hs || n < 1) return fail(-1, "bad argument");
    bool serial = (n == 1);
    for (int i = 0; i < n; ++i) { if (!hs[i]) return fail(-1, "null handle"); if (phased(hs[i]) && hs[i]->tkind == TGT_HOST) serial = true;
        }
    if (serial) {
        int worst = 0;
        for (int i = 0; i < n; ++i) { int rc = mcmcx_run(hs[i], upto); if (rc < 0) return rc; worst = std::max(worst, rc); }
        return worst;
    }
    std::vector<int> rcs(n, 0); std::vector<std::string> errs(n);
    std::vector<std::thread> th;
    for (int i = 1; i < n; ++i) th.emplace_back([&, i]() { rcs[i] = mcmcx_run(hs[i], upto); if (rcs[i] < 0) errs[i] = g_err; });
    rcs[0] = mcmcx_run(hs[0], upto); if (rcs[0] < 0) errs[0] = g_err;
    for (auto &t : th) t.join();
    int worst = 0;
    for (int i = 0; i < n; ++i) { if (rcs[i] < 0) return fail(rcs[i], "engine " + std::to_string(i) + ": " + errs[i]);
        worst = std::max(worst, rcs[i]); }
    return worst;
}

} // extern "C"

extern "C" {
// ------------------------------------------------------------------ debug probes (tests only)
// The kernel-selection tables (launch_step / launch_group / launch_scam): entry `index` as "family:name" into buf; returns the number of
// entries (so index = -1 with buf = NULL just counts).  Needs no device.
int mcmcx_debug_kernel_table(int32_t index, char *buf, int32_t len)
{
    const KernelEntry *tabs[] = {STEP_TABLE, GROUP_TABLE, SCAM_TABLE};
    const size_t ns[] = {sizeof(STEP_TABLE) / sizeof(STEP_TABLE[0]), sizeof(GROUP_TABLE) / sizeof(GROUP_TABLE[0]),
        sizeof(SCAM_TABLE) / sizeof(SCAM_TABLE[0])};
    int total = 0, k = index;
    for (int t = 0; t < 3; ++t) {
        if (k >= 0 && k < (int)ns[t] && buf && len > 0) { snprintf(buf, (size_t)len, "%s:%s", tabs[t][k].family, tabs[t][k].name);
            k = -1 - total - (int)ns[t]; }
        else if (k >= 0) k -= (int)ns[t];
        total += (int)ns[t];
    }
    return total;
}

int mcmcx_debug_math(int32_t op, int32_t n, const double *a, const double *b, double *out)
{
    if (n < 1 || !a || !out) return fail(-1, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(-10, "no HIP device: the mcmcx engine has no CPU fallback");
    DevBufs g;
    double *da = nullptr, *db = nullptr, *dout = nullptr;
    HIPCHK(g.alloc(&da, (size_t)n * 8)); HIPCHK(g.alloc(&dout, (size_t)n * 8));
    HIPCHK(hipMemcpy(da, a, (size_t)n * 8, hipMemcpyHostToDevice));
    if (b) { HIPCHK(g.alloc(&db, (size_t)n * 8)); HIPCHK(hipMemcpy(db, b, (size_t)n * 8, hipMemcpyHostToDevice)); }
    hipLaunchKernelGGL(debug_math_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, op, n, da, db, dout);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, dout, (size_t)n * 8, hipMemcpyDeviceToHost));
    return 0;
}

// Replace the SVD proposal factor of EVERY chain (per-chain mode, condmax > 0): R column-major d x d (scam: the rotation U),
// qcovstd (scam; may be NULL).  Lets a test feed the factors an external LAPACK returned at an adaptation
// (tests/test_gpu_parity.py: the MKL-linked reference's logged dgesvd results at d = 200).
int mcmcx_debug_set_factor(mcmcx_handle h, const double *R_colmajor, const double *qcovstd)
{
    if (!h || !h->inited || !R_colmajor) return fail(-1, "mcmcx_debug_set_factor: bad argument");
    if (!h->usesvd || h->pooled
        || !h->E.Rf) return fail(-45, "mcmcx_debug_set_factor: per-chain SVD factor only (condmax > 0, not pooled)");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<double> r(R_colmajor, R_colmajor + (size_t)h->d * h->d);
    int rc = dev_bcast(h, h->E.Rf, r); if (rc) return rc;
    if (qcovstd && h->E.qstd) { std::vector<double> q(qcovstd, qcovstd + h->d); if ((rc = dev_bcast(h, h->E.qstd, q))) return rc; }
    return 0;
}

int mcmcx_debug_rng(uint32_t seed, uint32_t chain_id, int32_t kind, int32_t n, double a, double b, double *out, uint64_t *nused)
{
    if (n < 1 || !out) return fail(-1, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(-10, "no HIP device: the mcmcx engine has no CPU fallback");
    DevBufs g;
    double *dout = nullptr; uint64_t *dn = nullptr;
    HIPCHK(g.alloc(&dout, (size_t)n * 8)); HIPCHK(g.alloc(&dn, 8));
    hipLaunchKernelGGL(debug_rng_kernel, dim3(1), dim3(64), 0, 0, seed, chain_id, kind, n, a, b, dout, dn);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, dout, (size_t)n * 8, hipMemcpyDeviceToHost));
    if (nused) HIPCHK(hipMemcpy(nused, dn, 8, hipMemcpyDeviceToHost));
    return 0;
}

} // extern "C"
