// mcx_host_launch.hpp -- which sampling kernel runs: cover predicates, the three selection tables (step / group / scam) and their
// launchers, the shared tables' uploads.
// Part of the ONE translation unit mcx_api.hip (included there, in this order: mcx_host_engine, mcx_host_linalg, mcx_host_launch,
// mcx_host_adapt, mcx_host_pooled, mcx_host_callbacks); not a stand-alone header.

// LDS of pooled_mfma_kernel: the tile's vector [d4][64] (+ the products [16 nt][64] when they need more than one pass)
// and the partial ss chains [4 nt][64]
static size_t pooled_mfma_lds(int d)
{
    const size_t d4 = (size_t)((d + 3) & ~3), nt = (size_t)((d + 15) / 16);
    // single pass: products and ss chains reuse the vector's rows
    const size_t rows = (nt <= 4) ? std::max(d4, 4 * nt) : d4 + 16 * nt + 4 * nt;
    return rows * 64 * sizeof(double);
}
static bool pooled_use_mfma(const mcmcx_engine *h)
{
    // (SCAM has its own)
    if (!h->pooled || (h->cfg.method != MCMCX_METHOD_DRAM && h->cfg.method != MCMCX_METHOD_RAM
        && h->cfg.method != MCMCX_METHOD_ER)) return false;
    // DR: with its dense tables, and above
    if (h->dodr) {
        // npar 20 (8.3e8 against 8.9e8 iterations/s
        if (h->cfg.method != MCMCX_METHOD_DRAM || !h->d_sharedR2T) return false;
        // for the lane kernel with its LDS vectors at 20;
        int dmin = 21;
        // 32: 5.6e8 / 2.5e8, 50: 3.2e8 / 0.8e8)
        if (h->sw.pooled_mfma_dr_min >= 0) dmin = h->sw.pooled_mfma_dr_min;
        if (h->d < dmin) return false;
    }
    if (h->sw.pooled_scalar > 0) return false;                                         // A/B switch for tests: the lane-per-chain kernel
    return pooled_mfma_lds(h->d) <= 160 * 1024;
}
// iteration cut at the evaluations
static bool phased(const mcmcx_engine *h) { return h->tkind == TGT_HOST || h->tkind == TGT_EXPCOLS || h->tkind == TGT_MODULE; }
// ... except where the DEVICE evaluates between the phases (the response-column target): the phases fused into one launch per segment
// (step_kernel_cols); MCMCX_COLS_PHASED=1 keeps the separate launches (A/B, tests)
static bool fused_cols(const mcmcx_engine *h) { return h->tkind == TGT_EXPCOLS && !(h->sw.cols_phased > 0); }
static bool phase_cut(const mcmcx_engine *h) { return phased(h) && !fused_cols(h); }
static size_t lds_bytes(const mcmcx_engine *h) { return (size_t)h->d * 64 * sizeof(double) * 2; }   // adapt / DR work vectors
static bool dr_fits_lds(const mcmcx_engine *h) { return lds_bytes(h) <= 160 * 1024; }          // npar <= 160
static size_t lds_step(const mcmcx_engine *h) { return (h->dodr && dr_fits_lds(h)) ? lds_bytes(h) : 0; }
// step_kernel_dr keeps the second stage's two vectors in LDS, step_kernel_dr_big in global scratch.  LDS pays while eight waves
// still fit a CU (npar <= 20: 6.9e8 against 6.0e8 iterations/s at 20); beyond, the waves it costs are worth more than the bytes
// it saves (npar 24: 3.1e8 against 4.1e8, 64: 3.2e7 / 4.6e7, 100: 0.8e7 / 1.7e7 -- tools/dr_sweep.py)
// (the pooled form, whose factors come through the scalar cache, is bound by that latency rather than by waves: LDS down to four
//  waves per CU -- npar 32: 2.5e8 against 2.2e8, 50: 7.4e7 / 8.1e7, 100: 0.8e7 / 1.3e7 -- tools/pooled_dr_probe.py)
static bool dr_vectors_in_lds(const mcmcx_engine *h, int min_waves = 8)
{
    if (!dr_fits_lds(h)) return false;
    if (h->sw.dr_big >= 0) return h->sw.dr_big == 0;                                  // A/B switch for tests: 1 = global scratch, 0 = LDS
    return lds_bytes(h) * (size_t)min_waves <= 160 * 1024;
}
static void launch_init(mcmcx_engine *h)
{ hipLaunchKernelGGL(init_kernel, dim3(h->ntiles), dim3(64), 0, h->stream, h->E); }
// ---- which sampling kernel runs: ONE table per launcher, walked in order -- the first entry whose predicate holds is launched and
// its name noted for mcmcx_last_kernel (bench.py labels its roofline with it).  The tables are exported through
// mcmcx_debug_kernel_table, so that tests/test_kernel_table.py can list every selectable instance and require a parity test that
// asserted each of them.  A configuration no entry accepts is an error of the run (launch_err), never a silent no-launch.
struct KernelEntry {
    const char *family;                                   // "step" (launch_step), "group" (launch_group), "scam" (launch_scam)
    const char *name;
    bool (*when)(const mcmcx_engine *);
    void (*launch)(mcmcx_engine *, int it0, int it1);
};
static void walk_table(mcmcx_engine *h, const KernelEntry *tab, size_t n, int it0, int it1)
{
    for (size_t i = 0; i < n; ++i)
        if (tab[i].when(h)) { h->last_kernel = tab[i].name; tab[i].launch(h, it0, it1); return; }
    h->launch_err = std::string("no ") + (n ? tab[0].family : "?") + " kernel covers this configuration";
}
// ---- the lane-group step kernel (mcx_group.hpp): four chains per wave, factors in registers
static const int GROUP_MAXSEG = 256;                  // iterations per launch (one accept byte per chain and iteration in d_accb)
static const int GROUP_MAX_NPAR = 64, GROUP_MAX_NPAR_DR = 32;      // (with delayed rejection three tables must fit a lane's registers)
// what the kernel covers: MCMC_run with per-chain Cholesky factors (method 'dram', with or without delayed rejection), one of
// the single-launch device targets, one response column
static bool group_covers(const mcmcx_engine *h)
{
    const mcmcx_config &c = h->cfg;
    return !h->pooled && (c.method == MCMCX_METHOD_DRAM || (c.method == MCMCX_METHOD_ER && !h->dodr)) && !h->usesvd && !phased(h)
        && h->ny == 1 &&
           (h->tkind == TGT_GAUSS || h->tkind == TGT_BANANA || h->tkind == TGT_EXPDATA) && h->d <= (h->dodr ? GROUP_MAX_NPAR_DR
               : GROUP_MAX_NPAR) &&
           !(h->tkind == TGT_BANANA && h->d < 2) && !(h->tkind == TGT_EXPDATA && h->d < 2);
}
// ... and where it is the faster one (tools/group_sweep.py, profiles/r04_a/group_sweep.txt: proposals/s of both kernel families over npar,
// target, delayed rejection and chain count).  Up to 16384 chains always: the chip is not full, a chain's iteration is latency, and
// sixteen lanes per chain with the factors on chip take 2-5 us where a lane takes 7-160 (4x-33x).  With the chip full, from npar 11
// on: 1.04-1.3x without delayed rejection up to npar 20 and 1.7-3x above, 1.1-3.7x with it; at npar <= 10 the lane kernels, which keep
// the factor in LDS there, stay ahead (group: 0.35-0.9x).
// ... and with which group width: four lanes per chain (sixteen chains per wave) for small npar with the chip full, where sixteen lanes
// would mostly idle (tools/quad_sweep.py, profiles/r04_b/quad_sweep.txt: without delayed rejection 1.3-3.6x the sixteen-lane form at npar
// <= 16 and 1.1-3.2x the lane kernels up to 131072 chains -- 1.5-1.7x at any count from npar 11 on; with it at npar <= 8)
static int group_width(const mcmcx_engine *h)
{
    if ((long long)h->cfg.nchains <= 16384 || h->d > 16 || h->cfg.updatesigma) return 16;
    if (!h->dodr) return 4;
    return h->d <= 8 ? 4 : 16;
}
static bool group_wins(const mcmcx_engine *h, int drm, int gw)
{
    const long long n = h->cfg.nchains;
    const int d = h->d;
    if (h->cfg.updatesigma) {
        // MCMC_updatesigma2's gamma sampler is a serial, data-dependent sequence of draws per chain: a group wave runs it for four chains,
        // a lane wave for 64 (tools/group_probe2.py: 2.3-12x up to 1024 chains, 0.9-5x at 16384, 0.45-0.8x beyond without delayed
        // rejection, 1.3-1.6x with it at npar 20)
        if (n <= 8192) return true;
        if (n <= 16384) return d >= 4;
        return h->dodr && d >= 11;
    }
    if (n <= 16384) return true;
    if (gw == 4) return h->dodr ? (drm == 2 || n <= 131072) : (d >= 11 || n <= 131072);
    return d >= 11 || (h->dodr && d >= 9 && n <= 131072);
}
// one instantiation per (group width, npar rounded up, delayed-rejection form, target kind); DRM 0 = none, 1 = the general form (R, R2, iC
// in registers), 2 = drscale a power of two (no R2; iC in LDS): the instantiation without R2 runs unless the device flag says that some
// factor leaves the range in which R'z / drscale is R2'z bit for bit; the general one is queued behind the same flag and returns at once
// otherwise
template <int GW, int D4, int DRM, int TK>
static void launch_group_inst(mcmcx_engine *h, int it0, int it1)
{
    const dim3 g(h->ntiles * (GW == 16 ? 16 : 4)), b(64);           // 64 / GW chains per wave
    const double *lam = h->E.tgt.lamT;
    if constexpr (DRM != 0 && D4 > GROUP_MAX_NPAR_DR) h->launch_err = "group kernel: no delayed-rejection instantiation above npar " +
        std::to_string(GROUP_MAX_NPAR_DR);
    else if constexpr (TK == TGT_EXPDATA && D4 != 4) h->launch_err = "group kernel: the expdata target has two parameters";
    else if constexpr (DRM == 0) hipLaunchKernelGGL((group_step_kernel<GW, D4, 0, TK>), g, b, 0, h->stream, h->E, it0, it1, lam, h->d_accb,
        (const int *)nullptr, 0);
    else if constexpr (DRM == 1) hipLaunchKernelGGL((group_step_kernel<GW, D4, 1, -1>), g, b, 0, h->stream, h->E, it0, it1, lam, h->d_accb,
        (const int *)nullptr, 0);
    else {
        hipLaunchKernelGGL((group_step_kernel<GW, D4, 2, TK>), g, b, 0, h->stream, h->E, it0, it1, lam, h->d_accb, (const int *)h->d_gflag,
            0);
        hipLaunchKernelGGL((group_step_kernel<GW, D4, 1, -1>), g, b, 0, h->stream, h->E, it0, it1, lam, h->d_accb, (const int *)h->d_gflag,
            1);
    }
}
template <int GW, int D4, int DRM>
static void launch_group_tk(mcmcx_engine *h, int it0, int it1)
{
    if (h->tkind == TGT_BANANA) launch_group_inst<GW, D4, DRM, TGT_BANANA>(h, it0, it1);
    else if (h->tkind == TGT_EXPDATA) launch_group_inst<GW, D4, DRM, TGT_EXPDATA>(h, it0, it1);
    else launch_group_inst<GW, D4, DRM, TGT_GAUSS>(h, it0, it1);
}
template <int GW, int DRM>
static void launch_group_d4(mcmcx_engine *h, int it0, int it1)
{
    if constexpr (GW == 4) {                              // quads: npar <= 16
        switch (h->group_d4) {
        case 4: launch_group_tk<4, 4, DRM>(h, it0, it1); break;
        case 8: launch_group_tk<4, 8, DRM>(h, it0, it1); break;
        case 12: launch_group_tk<4, 12, DRM>(h, it0, it1); break;
        case 16: launch_group_tk<4, 16, DRM>(h, it0, it1); break;
        default: h->launch_err = "group kernel (quads): npar > 16";
        }
    } else {
        switch (h->group_d4) {
        case 4: launch_group_tk<16, 4, DRM>(h, it0, it1); break;
        case 8: launch_group_tk<16, 8, DRM>(h, it0, it1); break;
        case 12: launch_group_tk<16, 12, DRM>(h, it0, it1); break;
        case 16: launch_group_tk<16, 16, DRM>(h, it0, it1); break;
        case 20: launch_group_tk<16, 20, DRM>(h, it0, it1); break;
        case 24: launch_group_tk<16, 24, DRM>(h, it0, it1); break;
        default:
            // DRM = 2 (iC as a square in LDS) is the engine's choice up to npar 24 only (mcmcx_init): larger sizes are not instantiated --
            // at 28 / 32 they could not hold the two waves per SIMD they would declare (Weak 12)
            if constexpr (DRM != 2) {
                switch (h->group_d4) {
                case 28: launch_group_tk<16, 28, DRM>(h, it0, it1); return;
                case 32: launch_group_tk<16, 32, DRM>(h, it0, it1); return;
                default: break;
                }
            }
            if constexpr (DRM == 0) {                                    // (above 32: sizes of eight, no delayed rejection)
                switch (h->group_d4) {
                case 40: launch_group_tk<16, 40, DRM>(h, it0, it1); return;
                case 48: launch_group_tk<16, 48, DRM>(h, it0, it1); return;
                case 56: launch_group_tk<16, 56, DRM>(h, it0, it1); return;
                case 64: launch_group_tk<16, 64, DRM>(h, it0, it1); return;
                default: break;
                }
            }
            h->launch_err = "group kernel: no instantiation for npar " + std::to_string(h->d) + " with delayed-rejection mode " +
                std::to_string(DRM);
        }
    }
}
static const KernelEntry GROUP_TABLE[] = {
    {"group", "group_step_kernel",            [](const mcmcx_engine *h) { return h->group_gw == 16 && h->group_drm == 0; },
        launch_group_d4<16, 0>},
    {"group", "group_step_kernel<DR>",        [](const mcmcx_engine *h) { return h->group_gw == 16 && h->group_drm == 1; },
        launch_group_d4<16, 1>},
    {"group", "group_step_kernel<DR2>",       [](const mcmcx_engine *h) { return h->group_gw == 16 && h->group_drm == 2; },
        launch_group_d4<16, 2>},
    {"group", "group_step_kernel<quad>",      [](const mcmcx_engine *h) { return h->group_gw == 4 && h->group_drm == 0; },
        launch_group_d4<4, 0>},
    {"group", "group_step_kernel<quad, DR>",  [](const mcmcx_engine *h) { return h->group_gw == 4 && h->group_drm == 1; },
        launch_group_d4<4, 1>},
    {"group", "group_step_kernel<quad, DR2>", [](const mcmcx_engine *h) { return h->group_gw == 4 && h->group_drm == 2; },
        launch_group_d4<4, 2>},
};
static void launch_group(mcmcx_engine *h, int it0, int it1)
{
    if (h->group_drm == 2 && h->group_check_due) {       // the factors have been rewritten since the last look
        (void)hipMemsetAsync(h->d_gflag, 0, sizeof(int), h->stream);
        hipLaunchKernelGGL(group_check_kernel, dim3(h->ntiles), dim3(64), 0, h->stream, h->E, h->d_gflag);
        h->group_check_due = false;
    }
    walk_table(h, GROUP_TABLE, sizeof(GROUP_TABLE) / sizeof(GROUP_TABLE[0]), it0, it1);
    if (h->d_accb && h->launch_err.empty()) {
        const long long n = (long long)(it1 - it0 + 1) * h->ntiles;
        hipLaunchKernelGGL(group_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->E, h->d_accb, it0, it1);
    }
}
// pooled_mfma_kernel<false, true>: two waves per SIMD (256 registers, some state spilled) pay with more tiles than SIMDs: from two per SIMD
// where the LDS vector lets eight waves on a CU (npar <= 39: +25 .. +55 %), from eight where it lets six (npar 50: 0.80 at 2048 tiles, 0.99
// at 4096, 1.14 at 16384); with one tile per SIMD the spills are all they buy (0.83 .. 0.90) -- tools/pooled_waves_probe.py
static bool pooled_two_waves(const mcmcx_engine *h)
{
    // (test switch: either instance on a small problem)
    if (h->sw.pooled_waves == 1 || h->sw.pooled_waves == 2) return h->sw.pooled_waves == 2;
    return h->ntiles >= (pooled_mfma_lds(h->d) * 8 <= 160 * 1024 ? 2048 : 8192);
}
// pooled_mfma_ks_kernel: the forty-row LDS vector (eight tiles on a CU whatever npar is) where the whole vector leaves fewer, for the
// problems that fill those CUs: npar 41..64, no delayed rejection, and two tiles per SIMD or more.  Against pooled_mfma_kernel<false, true>
// at 16384 tiles (tools/pooled_ks_sweep.py, profiles/r06_j): 1.00 at npar 41 / 44 (seven tiles there), 1.06 .. 1.10 at 45..52 (six),
// 1.12 .. 1.25 at 53..64 (five); with a non-Gaussian target 1.05 .. 1.42.  Against what the engine took before at fewer tiles
// (tools/pooled_ks_tiles.py; pooled_mfma_kernel<false> below 8192): 1.15 .. 1.22 at 2048 / 4096 tiles for npar 45 / 50, 0.98 / 1.02 at 64;
// 0.63 .. 0.69 at 1024 tiles (one per SIMD: the spills are all two waves buy there).
static bool pooled_forty_rows(const mcmcx_engine *h)
{
    const int d4 = (h->d + 3) & ~3;
    if (!pooled_use_mfma(h) || h->dodr || d4 <= PKS || d4 > 64) return false;
    // (test switch: either form on a small problem)
    if (h->sw.pooled_ks == 0 || h->sw.pooled_ks == 1) return h->sw.pooled_ks == 1;
    return h->sw.pooled_waves != 1 && h->ntiles >= 2048;
}
#define STEP_ARGS h->stream, h->E, it0, it1
#define STEP_RS (h->d_ramscale + it0)
#define STEP_TGT h->E.tgt.mu, h->E.tgt.lamT
#define G1 dim3(h->ntiles), dim3(64)
// method = 'ram' with few chains: sixteen lanes per chain, the factor in registers, dchud / dchdd on it there (mcx_group_ram.hpp) where it
// is the faster one (tools/ram_group_sweep.py, profiles/r05_b/ram_group_sweep.txt: chain-iterations/s of both families over npar, chain
// count and regime): one wave per SIMD and four chains per wave, so the chip holds 4096 chains at once and the kernel saturates there
// (1.68e8 / 4.6e8 / 1.1e9 chain-iterations/s at npar 50 / 20 / 10) -- 9x / 6x / 6x the lane kernels up to 4096 chains, still 2.9x / 1.7x /
// 2.0x at 16384 and 1.6x / 1.1x / 1.2x at 32768; from 65536 chains on the streaming kernels are ahead (0.93 / 0.74 / 0.63)
static bool ram_group_wins(const mcmcx_engine *h)
{
    const long long n = h->cfg.nchains;
    return n <= 16384 || (n <= 32768 && h->d >= 17);
}
static bool ram_group_covers(const mcmcx_engine *h)
{
    return !h->pooled && h->cfg.method == MCMCX_METHOD_RAM && !h->usesvd && !phased(h) && h->ny == 1 && h->d <= 64 &&
           (h->tkind == TGT_GAUSS || h->tkind == TGT_BANANA || h->tkind == TGT_EXPDATA) && !(h->tkind == TGT_BANANA && h->d < 2);
}
template <int D4>
static void launch_group_ram_d4(mcmcx_engine *h, int it0, int it1)
{
    hipLaunchKernelGGL((group_ram_kernel<D4, -1>), dim3(h->ntiles * 16), dim3(64), 0, h->stream, h->E, it0, it1,
        (const double *)h->d_ramscale, h->E.tgt.lamT, h->d_accb);
}
static void launch_group_ram(mcmcx_engine *h, int it0, int it1)
{
    switch (h->ram_group_d4) {
    case 16: launch_group_ram_d4<16>(h, it0, it1); break;
    case 32: launch_group_ram_d4<32>(h, it0, it1); break;
    case 56: launch_group_ram_d4<56>(h, it0, it1); break;
    case 64: launch_group_ram_d4<64>(h, it0, it1); break;
    default: h->launch_err = "group_ram_kernel: no instantiation for npar " + std::to_string(h->d); return;
    }
    if (h->d_accb) {
        const long long n = (long long)(it1 - it0 + 1) * h->ntiles;
        hipLaunchKernelGGL(group_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->E, h->d_accb, it0, it1);
    }
}
#ifdef MCX_VARIANTS                // measured negatives, built only by tools/build_variant.sh -DMCX_VARIANTS -- never part of libmcmcx.so
#include "../../tools/variants/variants.inc"
#else
#define MCX_VARIANT_STEP_ENTRIES
#define MCX_VARIANT_SVD_SWEEP(h, lss) false
#define MCX_VARIANT_COV(h, g8, n10, noff, it, mode) false
#endif
static const KernelEntry STEP_TABLE[] = {
    // ---- a device target with response columns (nycol >= 1 sums of squares per point): the phases of an iteration in one launch
    // (one instantiation per method class: MCMC_run_ram's carries the rank-one update's panels, the others do not)
    {"step", "step_kernel_cols<ram>", [](const mcmcx_engine *h) { return fused_cols(h) && h->E.method == M_RAM; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_cols<1>, G1, lds_step(h), STEP_ARGS,
         (const double *)h->d_ramscale, (const double *)nullptr, (const double *)nullptr, (const double *)nullptr); }},
    {"step", "step_kernel_cols", fused_cols,
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_cols<0>, G1, lds_step(h), STEP_ARGS,
         (const double *)h->d_ramscale, (const double *)(h->pooled ? h->E.sharedR : nullptr), (const double *)h->d_sharedR2,
         (const double *)h->d_sharediC); }},
    // ---- pooled mode (one shared factor)
    {"step", "pooled_mfma_kernel<true>", [](const mcmcx_engine *h) { return pooled_use_mfma(h) && h->dodr != 0; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(pooled_mfma_kernel<true>, G1, pooled_mfma_lds(h->d), STEP_ARGS, STEP_TGT,
         h->d_sharedRT, h->d_sharedR2T, h->d_sharediCd); }},
    MCX_VARIANT_STEP_ENTRIES
    {"step", "pooled_mfma_ks_kernel", pooled_forty_rows,
     [](mcmcx_engine *h, int it0, int it1) {
         // (npar 49..52: the fourth output block has four rows and goes through the 4 x 4 x 4 instruction)
         const size_t lds = (size_t)PKS * 64 * sizeof(double);
         if (((h->d + 3) & ~3) == 52) hipLaunchKernelGGL(pooled_mfma_ks_kernel<true>, G1, lds, STEP_ARGS, STEP_TGT, h->d_sharedRT);
         else hipLaunchKernelGGL(pooled_mfma_ks_kernel<false>, G1, lds, STEP_ARGS, STEP_TGT, h->d_sharedRT); }},
    {"step", "pooled_mfma_kernel<false, true>", [](const mcmcx_engine *h) { return pooled_use_mfma(h) && pooled_two_waves(h); },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL((pooled_mfma_kernel<false, true>), G1, pooled_mfma_lds(h->d), STEP_ARGS,
         STEP_TGT, h->d_sharedRT, (const double *)nullptr, (const double *)nullptr); }},
    {"step", "pooled_mfma_kernel<false>", [](const mcmcx_engine *h) { return pooled_use_mfma(h); },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(pooled_mfma_kernel<false>, G1, pooled_mfma_lds(h->d), STEP_ARGS, STEP_TGT,
         h->d_sharedRT, (const double *)nullptr, (const double *)nullptr); }},
    {"step", "step_kernel_pooled_dr_big", [](const mcmcx_engine *h) { return h->pooled && h->dodr && !dr_vectors_in_lds(h, 4); },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_pooled_dr_big, G1, 0, STEP_ARGS, STEP_RS, STEP_TGT,
         h->E.sharedR, h->d_sharedR2, h->d_sharediC); }},
    {"step", "step_kernel_pooled_dr", [](const mcmcx_engine *h) { return h->pooled && h->dodr; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_pooled_dr, G1, lds_step(h), STEP_ARGS, STEP_RS, STEP_TGT,
         h->E.sharedR, h->d_sharedR2, h->d_sharediC); }},
    {"step", "step_kernel<false, false, true>", [](const mcmcx_engine *h) { return h->pooled != 0; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL((step_kernel<false, false, true>), G1, 0, STEP_ARGS, STEP_RS, STEP_TGT,
         h->E.sharedR); }},
    // ---- method = 'ram', per-chain factors
    {"step", "group_ram_kernel", [](const mcmcx_engine *h) { return h->ram_group_d4 != 0; }, launch_group_ram},
    {"step", "step_kernel_ram_fullr", [](const mcmcx_engine *h) { return h->E.method == M_RAM && h->usesvd; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_ram_fullr, G1, 0, STEP_ARGS, STEP_RS, STEP_TGT, h->E.sharedR);
         }},
    {"step", "step_kernel_ram_ldsr", [](const mcmcx_engine *h) { return h->E.method == M_RAM && h->E.lds_scratch == 3; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_ram_ldsr, G1, (size_t)(2 * h->d + h->P) * 64 * sizeof(double),
         STEP_ARGS, STEP_RS, STEP_TGT, h->E.sharedR); }},
    {"step", "step_kernel_ram_wide", [](const mcmcx_engine *h) {
        return h->E.method == M_RAM && h->d > RAM_SMALL_MAX && h->sw.ram_wide != 0; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_ram_wide, G1, (size_t)NLC * 2 * 64 * sizeof(double), STEP_ARGS,
         STEP_RS, STEP_TGT, h->E.sharedR); }},
    {"step", "step_kernel<true, false, false>", [](const mcmcx_engine *h) { return h->E.method == M_RAM; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL((step_kernel<true, false, false>), G1,
         (size_t)NLC * 2 * 64 * sizeof(double), STEP_ARGS, STEP_RS, STEP_TGT, h->E.sharedR); }},
    // ---- delayed rejection, per-chain factors: the second stage's two vectors in global scratch / in LDS
    {"step", "step_kernel_dr_big", [](const mcmcx_engine *h) { return h->dodr && !dr_vectors_in_lds(h); },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_dr_big, G1, 0, STEP_ARGS, STEP_TGT); }},
    {"step", "step_kernel_dr", [](const mcmcx_engine *h) { return h->dodr != 0; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_dr, G1, lds_step(h), STEP_ARGS, STEP_TGT); }},
    // ---- AM / Metropolis / early rejection: state + factor in LDS, state in LDS, nothing in LDS
    {"step", "step_kernel_ldsr", [](const mcmcx_engine *h) { return h->E.lds_scratch == 2; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_ldsr, G1, (size_t)(2 * h->d + h->P) * 64 * sizeof(double),
         STEP_ARGS, STEP_RS, STEP_TGT, h->E.sharedR); }},
    {"step", "step_kernel_ldsv", [](const mcmcx_engine *h) { return h->E.lds_scratch != 0; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_ldsv, G1, (size_t)4 * h->d * 64 * sizeof(double), STEP_ARGS,
         STEP_RS, STEP_TGT, h->E.sharedR); }},
    {"step", "step_kernel<false, false, false>", [](const mcmcx_engine *) { return true; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL((step_kernel<false, false, false>), G1, 0, STEP_ARGS, STEP_RS, STEP_TGT,
         h->E.sharedR); }},
};
static void launch_step(mcmcx_engine *h, int it0, int it1)
{
    if (h->group_d4) { launch_group(h, it0, it1); return; }
    walk_table(h, STEP_TABLE, sizeof(STEP_TABLE) / sizeof(STEP_TABLE[0]), it0, it1);
}
// every chain's copy of a K-vector, filled on the device
static int dev_bcast(mcmcx_engine *h, double *dst, const std::vector<double> &v)
{
    double *tmp = nullptr;
    HIPCHK(hipMalloc(&tmp, v.size() * sizeof(double)));
    hipError_t e = hipMemcpy(tmp, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const unsigned gy = (unsigned)std::min<size_t>(v.size(), 64);
        hipLaunchKernelGGL(bcast_kernel, dim3(h->ntiles, gy), dim3(64), 0, h->stream, dst, tmp, v.size());
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    }
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(-100, hipGetErrorString(e));
    return 0;
}
static int upload_shared_rt(mcmcx_engine *h)
{
    const int d = h->d, d4 = (d + 3) & ~3;
    std::vector<double> m((size_t)d4 * d + PWS, 0.0);
    for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i) m[(size_t)i * d + j] = h->pool_R[h_pidx(i, j, d)];
    HIPCHK(hipMemcpyAsync(h->d_sharedRT, m.data(), m.size() * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
static int upload_shared_rf(mcmcx_engine *h)            // dense M[s*d + o] = Rf(o, s): the column-major factor as it stands, pad rows zero
{
    const int d = h->d, d4 = (d + 3) & ~3;
    std::vector<double> m((size_t)d4 * d + PWS, 0.0);
    memcpy(m.data(), h->pool_Rf.data(), (size_t)d * d * 8);
    HIPCHK(hipMemcpyAsync(h->d_sharedRT, m.data(), m.size() * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
// d4 rows (pad rows zero) + slack
static size_t shared_u_stride(const mcmcx_engine *h) { return (size_t)((h->d + 3) & ~3) * h->d + PWS; }
// X [16 nt][64], Q [4 nt][64], zb, fl, mu [16 nt]
static size_t scam_pooled_lds(int d) {
    return ((size_t)((d + 15) / 16) * (16 + 4) * 64 + 128 + (size_t)((d + 15) / 16) * 16) * sizeof(double); }
static int upload_shared_u(mcmcx_engine *h)
{
    if (h->scam_replicated) {                           // every chain's own copy of the one rotation and its scales
        int rc = dev_bcast(h, h->E.Rf, h->pool_U); if (rc) return rc;
        return dev_bcast(h, h->E.qstd, h->pool_std);
    }
    const int d = h->d; const size_t st = shared_u_stride(h);
    std::vector<double> b(2 * st + d, 0.0);
    for (int j = 0; j < d; ++j) for (int i = 0; i < d; ++i) { b[(size_t)j * d + i] = h->pool_U[(size_t)j * d + i];
        b[st + (size_t)i * d + j] = h->pool_U[(size_t)j * d + i]; }
    for (int i = 0; i < d; ++i) b[2 * st + i] = h->pool_std[i];
    HIPCHK(hipMemcpyAsync(h->d_sharedU, b.data(), b.size() * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
// 13..15 output blocks (npar 193..240): twelve waves of 170 registers (scam_pooled12_kernel) instead of sixteen of 128
static bool scam_use_12(const mcmcx_engine *h)
{
    const int nt = (h->d + 15) / 16;
    if (h->sw.scam_pooled_16 > 0) return false;                                       // A/B switch for tests: the sixteen-wave layout
    return nt >= 13 && nt <= 15;
}
// opt-in fast proposals with the Gaussian target: the workgroup-per-tile kernel of the pooled mode with each chain's own
// rotation column (g_U = nullptr) -- the target's d x d product on the matrix cores instead of lane by lane
static bool scam_fast_tile_kernel(const mcmcx_engine *h)
{
    return h->cfg.scam_fast && h->tkind == TGT_GAUSS && !h->has_lo && !h->has_hi && !h->has_pri && scam_pooled_lds(h->d) <= 160 * 1024 &&
           !(h->sw.scam_fast_lanes > 0);
}
// few tiles: several waves per tile (scam_mw_kernel), so that a sub-step is not bound by the latency of one wave's loads
// while most of the chip idles -- as many waves as keep the chip's ~2048 resident-wave slots busy, at most 8 (sixteen
// waves of 128 registers spill the products' panels)
static size_t scam_mw_lds(const mcmcx_engine *h) { return (size_t)(4 * ((h->d + 15) / 16) + 2) * 64 * sizeof(double); }
static int scam_tile_waves(const mcmcx_engine *h)
{
    int nw = 1;
    // (1024 tiles x npar 200: 8.4e6 at two waves per tile, 8.9e6 at four, 9.0e6 at eight -- profiles/r05_d/c5rep_waves.txt)
    while (nw < 8 && (long long)h->ntiles * nw * 2 <= 8192) nw *= 2;
    { const int v = h->sw.scam_waves; if (v == 1 || v == 2 || v == 4 || v == 8) nw = v; }   // A/B switch for tests
    // (npar > 1200: the one-wave kernel needs no LDS)
    if (scam_mw_lds(h) > 160 * 1024) nw = 1;
    return nw;
}
#define SCAM_POOLED_ARGS scam_pooled_lds(h->d), h->stream, h->E, it0, it1, h->E.tgt.mu, h->E.tgt.lamT
static const KernelEntry SCAM_TABLE[] = {
    {"scam", "step_kernel_cols<scam>", [](const mcmcx_engine *h) { return fused_cols(h) && !h->pooled; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_cols<2>, G1, 0, STEP_ARGS, (const double *)h->d_ramscale,
         (const double *)nullptr, (const double *)nullptr, (const double *)nullptr); }},
    // pooled: 16-row output blocks, min(12, 4*(nt/4)) block waves + 4 chain-group waves; twelve waves of 170 registers for 13..15 blocks
    {"scam", "scam_pooled12_kernel", [](const mcmcx_engine *h) { return h->pooled && !h->scam_replicated && scam_use_12(h); },
     [](mcmcx_engine *h, int it0, int it1) { const size_t st = shared_u_stride(h);
        hipLaunchKernelGGL(scam_pooled12_kernel, dim3(h->ntiles), dim3(768), SCAM_POOLED_ARGS, h->d_sharedU, h->d_sharedU + st,
            h->d_sharedU + 2 * st); }},
    {"scam", "scam_pooled_kernel", [](const mcmcx_engine *h) { return h->pooled && !h->scam_replicated; },
     [](mcmcx_engine *h, int it0, int it1) { const size_t st = shared_u_stride(h); const int nw = 4 + std::min(12, ((h->d + 15) / 16) & ~3);
        hipLaunchKernelGGL(scam_pooled_kernel, dim3(h->ntiles), dim3(64 * nw), SCAM_POOLED_ARGS, h->d_sharedU, h->d_sharedU + st,
            h->d_sharedU + 2 * st); }},
    // (the sixteen-wave layout whatever npar: every lane fetches the column of its own chain's factor per sub-step, and four
    //  waves per SIMD cover that better than three -- 4.62e8 against 4.45e8 proposals/s at npar 200)
    {"scam", "scam_pooled_kernel<per-chain>", scam_fast_tile_kernel,
     [](mcmcx_engine *h, int it0, int it1) { const int nw = 4 + std::min(12, ((h->d + 15) / 16) & ~3);
        hipLaunchKernelGGL(scam_pooled_kernel, dim3(h->ntiles), dim3(64 * nw), SCAM_POOLED_ARGS, (const double *)nullptr,
            (const double *)nullptr, (const double *)nullptr); }},
    {"scam", "scam_mw_kernel<8>", [](const mcmcx_engine *h) { return scam_tile_waves(h) == 8; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(scam_mw_kernel<8>, dim3(h->ntiles), dim3(512), scam_mw_lds(h), STEP_ARGS,
         STEP_TGT); }},
    {"scam", "scam_mw_kernel<4>", [](const mcmcx_engine *h) { return scam_tile_waves(h) == 4; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(scam_mw_kernel<4>, dim3(h->ntiles), dim3(256), scam_mw_lds(h), STEP_ARGS,
         STEP_TGT); }},
    {"scam", "scam_mw_kernel<2>", [](const mcmcx_engine *h) { return scam_tile_waves(h) == 2; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(scam_mw_kernel<2>, dim3(h->ntiles), dim3(128), scam_mw_lds(h), STEP_ARGS,
         STEP_TGT); }},
    {"scam", "scam_kernel", [](const mcmcx_engine *) { return true; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(scam_kernel, G1, 0, STEP_ARGS, STEP_TGT); }},
};
static void launch_scam(mcmcx_engine *h, int it0, int it1)
{
    walk_table(h, SCAM_TABLE, sizeof(SCAM_TABLE) / sizeof(SCAM_TABLE[0]), it0, it1);
}
