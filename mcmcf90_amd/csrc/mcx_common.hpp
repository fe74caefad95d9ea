// mcx_common.hpp -- what every kernel family shares: the engine's device view (EngineDev), the tile-interleaved layout (TIDX, GV ...),
// the device-resident targets (ssfunction / priorfun / checkbounds), the normal generator (normal_bm, mcmcrand.F90:166-190)
// (one of the family headers mcx_kernels.hpp includes, in this order: mcx_common, mcx_products, mcx_step, mcx_scam, mcx_pooled, mcx_phase,
// mcx_adapt, mcx_svd, mcx_moments)
#pragma once
#include "mcx_device.hpp"
#include <type_traits>

namespace mcx {


// EXPCOLS / MODULE: host side only (the device sees TGT_HOST + an evaluation kernel between the phases)
enum { TGT_GAUSS = 0, TGT_BANANA = 1, TGT_EXPDATA = 2, TGT_HOST = 3, TGT_EXPCOLS = 4, TGT_MODULE = 5 };
enum { M_DRAM = 0, M_RAM = 1, M_ER = 3 };

// per-chain scalar slots (doubles)
// S_WNEW: chainwsum after the blocked covariance update (adapt_cov_diag_kernel -> adapt_post_kernel)
enum { S_SS1 = 0, S_PRI1, S_SIGMA2, S_ALPHA12, S_SAVEDY, S_WSUM, S_WNEW, NSCAL };
// per-chain integer slots (u32)
enum { I_SAVED = 0, I_STAYED, I_BNDSTAYED, I_DRACC, I_DRTRIES, I_CHAININD, I_CURCOUNT, I_STATUS,
       // I_PDESC: 1 after a successful RAM downdate; I_DOWNS: RAM iterations with a < 0 (choldowndate)
       I_LASTFREQ, I_BASECNT, I_WINSTART, I_INFO, I_ERSTAYED, I_PDESC, I_DOWNS, I_ADFLAGS, I_NR, I_BSTART, NICTR };

// status bits
enum { ST_RAM_DOWNDATE_FAIL = 1, ST_CHOL_FAIL = 2, ST_POTRI_FAIL = 4 };

struct DevTarget {
    int kind;
    const double *mu, *lamT;    // gauss: mean[d] and the precision matrix transposed, lamT[j*d+i] = lam(i,j) (padded)
    double b;                   // banana
    int ndata;                  // expdata
    const double *x, *y;        // y: [ncols][ndata] for the response-column target
    int ncols;
    const double *lo, *hi;      // box bounds or nullptr
    const double *pmu, *psig;   // Gaussian priors or nullptr
};

struct EngineDev {
    int d, P, ntiles;
    int method, dodr, updatesigma, doadapt, doburnin, burnintime, greedy, adapthist;
    double initcmatn;
    double gam_shape;           // N0/2 + nobs/2           (MCMC_DRAM.F90:201)
    double N0S02;               // N0*S02
    double alphatarget, drscale, scalelimit, scalefactor;
    DevTarget tgt;
    // state, tile-interleaved
    double *theta, *cand, *zs, *cs, *scal, *R, *R2, *iC, *Rtmp;   // cand/zs [d], cs [2d]: per-chain scratch vectors
    // [2d] per chain: the two quadratic-form vectors of pooled delayed rejection when LDS would cost waves (step_kernel_pooled_dr_big)
    double *xscr;
    double *cmat, *mean, *basetheta;
    const double *cmat0p, *par0;    // packed upper cmat0 [P], par0 [d] (shared by all chains)
    uint32_t *ictr;
    uint64_t *rngn;
    uint32_t k0, chain_id0;
    // history ring: slot = it % wcap; hist[(tile*wcap + slot)*(d+1) + k][lane]
    int wcap, record_s2;
    double *hist, *s2hist;
    uint64_t *wacc;             // [tile*wcap + slot]
    uint64_t *accmask;          // [(it-1)*ntiles + tile] or nullptr
    uint64_t *rowlist;          // [(tile*(wcap+1) + r)*64 + lane]  (slot | weight<<32)
    const double *sharedR;      // pooled mode: the one packed factor all chains propose with
    // SVD paths (condmax > 0 / method='scam'): full column-major d x d factors per chain, element (i,j) at j*d+i
    int usesvd, doscam; double condmax;
    int scam_fast;                  // opt-in: componentwise proposals as theta + delta U(:,j) (mcmcx_config::scam_fast)
    double *Rf, *R2f, *qstd, *Gw, *Vw;
    // host-callback targets: per-chain evaluation results (inbounds, prior, ss) and state carried between phases
    double *hev, *hx;
    // response columns (nycol, mcmc.F90:30-33): ny > 1 only with host callbacks.  hs = d + ny doubles per history row
    // (theta, then ss per column); hev holds ny ss values per chain; per-chain vectors ssv (current ss), s2v (sigma2),
    // ss2v (first-stage ss kept for the DR formulas), gshapev[ny] = N0/2 + nobs(j)/2 (shared)
    int ny, hs;
    double *ssv, *s2v, *ss2v;
    const double *gshapev;
    // small npar, plain AM step kernel: the state vector and the per-chain scratch vectors (theta, candidate, two normal vectors)
    // live in LDS for the launch -- their store -> load chains are what an iteration waits for when the factor is small
    int lds_scratch;
    // delayed rejection: the two npar-vectors of the second stage live in LDS (1) or, where 2 x npar x 512 bytes do not fit a
    // CU's 160 KiB (npar > 160), in the chain's global scratch (0)
    int dr_lds;
};

#define TIDX(base, tile, K, k, lane) ((base)[((size_t)(tile) * (size_t)(K) + (size_t)(k)) * 64 + (lane)])

// Packed upper triangle, ROW-major: element (i,j), i <= j, sits at rowstart(i) + (j - i).
// Every sweep of the factor (proposal, update, downdate) walks whole rows, forwards or backwards,
// so a tile's factor is one sequential HBM stream of 512-byte wave segments.
MCX_DEV int rowstart(int i, int d) { return i * d - (i * (i - 1)) / 2; }
MCX_DEV int pidx(int i, int j, int d) { return rowstart(i, d) + (j - i); }

constexpr int CH = 8;     // row elements per batch; two batches (2 x 8 x 512 B) are in flight per wave

// The one per-lane d-vector of a wave lives in LDS as X[j*64 + lane] (conflict-free ds_read/write_b64).
#define XL(j) X[(j) * 64 + lane]
// element k of a tile-interleaved global vector whose tile base is `p` (uniform pointer)
#define GV(p, k) (p)[(size_t)(k) * 64 + lane]
#define GV2(p, k, c) (p)[(size_t)(k) * 64 + (c)]
// streaming (non-temporal) access for the factor, which is touched once per iteration and should not evict
// the small per-chain scratch vectors from L2 / Infinity Cache
#define LDNT(p, k) __builtin_nontemporal_load(&(p)[(size_t)(k) * 64 + lane])
#define STNT(p, k, v) __builtin_nontemporal_store((v), &(p)[(size_t)(k) * 64 + lane])
// the downdate's second sweep re-reads what the first one just streamed and writes partial segments (only the downdate
// lanes): plain accesses, so that L2 can serve the re-read and merge the partial stores (measured: +1..8 %)
#define LDB(p, k) GV(p, k)
#define STB(p, k, v) (GV(p, k) = (v))

// Software-pipelined sweep over elements k0..n-1 of one packed row (rowp[k], element stride 64):
// the next batch of CH elements is requested before the current one is consumed, and the
// ragged last batch is loaded with clamped addresses, so no load of a row is ever issued alone.
// f(k, r) is called for k ascending.
MCX_DEV void load_batch(double (&r)[CH], const double *rowp, int lane, int k, int n)
{
#pragma unroll
    for (int u = 0; u < CH; ++u) { int kk = k + u; kk = kk < n ? kk : n - 1; r[u] = GV(rowp, kk); }
}
template <typename F>
MCX_DEV void sweep(const double *rowp, int lane, int k0, int n, F &&f)
{
    double ra[CH], rb[CH];
    int k = k0;
    if (k < n) load_batch(ra, rowp, lane, k, n);
    while (k < n) {
        int k2 = k + CH;
        if (k2 < n) load_batch(rb, rowp, lane, k2, n);
        if (k2 <= n) {
#pragma unroll
            for (int u = 0; u < CH; ++u) f(k + u, ra[u]);
        } else {
#pragma unroll
            for (int u = 0; u < CH; ++u) if (k + u < n) f(k + u, ra[u]);
        }
        k = k2;
        if (k >= n) break;
        int k3 = k + CH;
        if (k3 < n) load_batch(ra, rowp, lane, k3, n);
        if (k3 <= n) {
#pragma unroll
            for (int u = 0; u < CH; ++u) f(k + u, rb[u]);
        } else {
#pragma unroll
            for (int u = 0; u < CH; ++u) if (k + u < n) f(k + u, rb[u]);
        }
        k = k3;
    }
}

// dst[k] = src[k] (and h[k] when h is given), k < d, eight elements' loads in flight: written element by element a copy waits
// for each load before its store and cannot issue the next load before that store (the compiler must assume that the vectors
// overlap) -- npar cache round trips in a row at every accepted move
MCX_DEV void copy_vec(double *dst, const double *src, double *h, int lane, int d)
{
    int k = 0;
    for (; k + 8 <= d; k += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = GV(src, k + u);
#pragma unroll
        for (int u = 0; u < 8; ++u) { GV(dst, k + u) = v[u]; if (h) GV(h, k + u) = v[u]; }
    }
    for (; k < d; ++k) { const double v = GV(src, k); GV(dst, k) = v; if (h) GV(h, k) = v; }
}

// copy_vec with NL loads in flight and no element-by-element tail (the last batch re-reads its last element): for a wave that has its SIMD
// almost to itself (pooled_mfma_kernel) every batch is a cache round trip nobody else covers
template <int NL>
MCX_DEV void copy_vec_wide(double *dst, const double *src, double *h, int lane, int d)
{
    for (int k = 0; k < d; k += NL) {
        double v[NL];
#pragma unroll
        for (int u = 0; u < NL; ++u) v[u] = GV(src, (k + u < d) ? k + u : d - 1);
#pragma unroll
        for (int u = 0; u < NL; ++u) if (k + u < d) { GV(dst, k + u) = v[u]; if (h) GV(h, k + u) = v[u]; }
    }
}

// dst[e] = f(src[e]), e < n, eight loads in flight (dst may be src): the element-by-element loop is a load-op-store round trip per
// element for the same reason as in copy_vec
template <typename F>
MCX_DEV void map_vec(double *dst, const double *src, int lane, int n, F &&f)
{
    int e = 0;
    for (; e + 8 <= n; e += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = GV(src, e + u);
#pragma unroll
        for (int u = 0; u < 8; ++u) GV(dst, e + u) = f(v[u]);
    }
    for (; e < n; ++e) GV(dst, e) = f(GV(src, e));
}

// The same sweep handing over a whole batch at a time: f(k, r, m) sees elements k..k+m-1 (m <= CH) in r[0..m-1], k ascending.
template <typename F>
MCX_DEV void sweep_batches(const double *rowp, int lane, int k0, int n, F &&f)
{
    double ra[CH], rb[CH];
    int k = k0;
    if (k < n) load_batch(ra, rowp, lane, k, n);
    while (k < n) {
        int k2 = k + CH;
        if (k2 < n) load_batch(rb, rowp, lane, k2, n);
        f(k, ra, (k2 <= n) ? CH : n - k);
        k = k2;
        if (k >= n) break;
        int k3 = k + CH;
        if (k3 < n) load_batch(ra, rowp, lane, k3, n);
        f(k, rb, (k3 <= n) ? CH : n - k);
        k = k3;
    }
}

// ---------------------------------------------------------------- targets (user ssfunction / priorfun / checkbounds)
// The candidate is read from a per-chain global scratch vector c_t (element stride 64); the
// Gaussian target works on 16x16 register panels: y[16] (rows) x v[16] (columns), precision matrix
// through scalar loads of its transpose (lamT[j*d + i] = lam(i,j), padded by PW doubles).
constexpr int PW = 8;     // panel width: columns (or rows) of per-lane state held in registers
#ifndef MCX_RW
#define MCX_RW 10
#endif
#ifndef MCX_TW
#define MCX_TW 10
#endif
constexpr int RW = MCX_RW;  // panel width of the RAM sweep (d = 50: five full panels)
// step_kernel_ram_wide (npar > RAM_SMALL_MAX, round 4): column panels up to RW_WIDE wide, as few as that allows and as equal as possible
// (npar 50: 17 + 17 + 16) -- every panel re-reads the rotations and the next normals of the rows above it.  A kernel of its own: the
// narrow panels in 17-element register rows, or both widths instantiated in one kernel, cost 11-15 % at npar <= 20 (tools/ram_rw_probe.py);
// 19 columns spill at two waves per SIMD.
#ifndef MCX_RW_WIDE
#define MCX_RW_WIDE 17
#endif
constexpr int RW_WIDE = MCX_RW_WIDE, RAM_SMALL_MAX = 20;
MCX_DEV int ram_panel_width(int d, int rwmax) { const int np = (d + rwmax - 1) / rwmax; return (d + np - 1) / np; }
constexpr int TW = MCX_TW;  // panel width of the per-chain triangular product

// One block of 16 rows of the Gaussian target (mcxt_ss_gauss, oracle/mcx_targets.h): y_i = sum_j lam(i,j) v_j as fma chains
// ascending in j, and the block's four partial chains q_k over the rows B0 + k + 4r.  ss is the running sum of the q_k over
// the blocks in order; blocks are independent of one another (a workgroup's waves share them in scam_mw_kernel).
MCX_DEV void gauss_block_q(int d, int lane, const double *c_t, const double *__restrict__ g_mu, const double *__restrict__ g_lamT,
                           int B0, double (&q)[4])
{
    q[0] = q[1] = q[2] = q[3] = 0.0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int I0 = B0 + PW * h;
        if (I0 < d) {
            const int nr = (d - I0) < PW ? (d - I0) : PW;
            double y[PW];
#pragma unroll
            for (int u = 0; u < PW; ++u) y[u] = 0.0;
            for (int J0 = 0; J0 < d; J0 += PW) {
                const int nc = (d - J0) < PW ? (d - J0) : PW;
                double v[PW];
#pragma unroll
                for (int w = 0; w < PW; ++w) { int j = J0 + (w < nc ? w : nc - 1); v[w] = GV(c_t, j) - g_mu[j]; }
#pragma unroll
                for (int w = 0; w < PW; ++w) {
                    if (w < nc) {
                        const double *__restrict__ lrow = g_lamT + (size_t)(J0 + w) * d + I0;
#pragma unroll
                        for (int u = 0; u < PW; ++u) y[u] = dfma(lrow[u], v[w], y[u]);
                    }
                }
            }
            double vi[PW];
#pragma unroll
            for (int u = 0; u < PW; ++u) { int i = I0 + (u < nr ? u : nr - 1); vi[u] = GV(c_t, i) - g_mu[i]; }
#pragma unroll
            for (int u = 0; u < PW; ++u) {
                if (u < nr) { if (h == 0 && u < 4) q[u & 3] = y[u] * vi[u]; else q[u & 3] = dfma(y[u], vi[u], q[u & 3]); }
            }
        }
    }
}

// WIDE: keep every row accumulator in registers and read the candidate once (pays when the kernel is
// bandwidth-bound: RAM); otherwise one row panel at a time (fewer registers: pooled / AM / DR kernels).
template <bool WIDE>
MCX_DEV double target_ss(const DevTarget &t, int d, int lane, const double *c_t,
                         const double *__restrict__ g_mu, const double *__restrict__ g_lamT)
{
    double ss = 0.0;
    // Gaussian: ss = v' Lam v in the order of mcxt_ss_gauss (oracle/mcx_targets.h): y_i = fma chain over j from 0; per
    // block of 16 rows four partial chains q_k over the rows 16t + k + 4r; ss = running sum of the q_k.
    if (WIDE && t.kind == TGT_GAUSS && d <= 8 * PW) {
        // Row accumulators of NPM panels of PW (32 rows) stay in registers while the columns stream by, so the candidate
        // is read once per 32 rows instead of once per row panel (all 64 rows at once spills).
        constexpr int NPM = 4;
        const int np = (d + PW - 1) / PW;
        for (int G0 = 0; G0 < np; G0 += NPM) {
            double y[NPM][PW];
#pragma unroll
            for (int p = 0; p < NPM; ++p)
#pragma unroll
                for (int u = 0; u < PW; ++u) y[p][u] = 0.0;
            for (int J0 = 0; J0 < d; J0 += PW) {
                const int nc = (d - J0) < PW ? (d - J0) : PW;
                double v[PW];
#pragma unroll
                for (int w = 0; w < PW; ++w) { int j = J0 + (w < nc ? w : nc - 1); v[w] = GV(c_t, j) - g_mu[j]; }
#pragma unroll
                for (int w = 0; w < PW; ++w) {
                    if (w < nc) {
                        const double *__restrict__ lcol = g_lamT + (size_t)(J0 + w) * d + (size_t)G0 * PW;     // lam(32 G0/4 .., J0+w)
#pragma unroll
                        for (int p = 0; p < NPM; ++p) {
                            if (G0 + p < np) {
#pragma unroll
                                for (int u = 0; u < PW; ++u) y[p][u] = dfma(lcol[p * PW + u], v[w], y[p][u]);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int tb = 0; tb < NPM / 2; ++tb) {
                if (G0 + 2 * tb < np) {
                    double q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int p = 2 * tb + h;
                        if (G0 + p < np) {
                            const int I0 = (G0 + p) * PW;
                            const int nr = (d - I0) < PW ? (d - I0) : PW;
                            double vi[PW];
#pragma unroll
                            for (int u = 0; u < PW; ++u) { int i = I0 + (u < nr ? u : nr - 1); vi[u] = GV(c_t, i) - g_mu[i]; }
#pragma unroll
                            for (int u = 0; u < PW; ++u) {
                                if (u < nr) { if (h == 0 && u < 4) q[u & 3] = y[p][u] * vi[u]; else q[u & 3] = dfma(y[p][u], vi[u],
                                    q[u & 3]); }
                            }
                        }
                    }
                    const int B0 = (G0 + 2 * tb) * PW;
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (B0 + k < d) ss = (B0 == 0 && k == 0) ? q[0] : ss + q[k];
                }
            }
        }
    } else if (t.kind == TGT_GAUSS) {
        for (int B0 = 0; B0 < d; B0 += 16) {
            double q[4];
            gauss_block_q(d, lane, c_t, g_mu, g_lamT, B0, q);
#pragma unroll
            for (int k = 0; k < 4; ++k) if (B0 + k < d) ss = (B0 == 0 && k == 0) ? q[0] : ss + q[k];
        }
    } else if (t.kind == TGT_BANANA) {
        double th0 = GV(c_t, 0), th1 = GV(c_t, 1);
        double t1 = th0 * th0;
        double q = dfma(t.b, t1, th1) - 100.0 * t.b;
        ss = dfma(q, q, t1 / 100.0);
#pragma unroll 4
        for (int k = 2; k < d; ++k) { double v = GV(c_t, k); ss = dfma(v, v, ss); }
    } else {
        double th0 = GV(c_t, 0), th1 = GV(c_t, 1);
        for (int i = 0; i < t.ndata; ++i) {
            double r = t.y[i] - th0 * d_exp(-(th1 * t.x[i]));
            ss = dfma(r, r, ss);
        }
    }
    return ss;
}

MCX_DEV double target_prior(const DevTarget &t, int d, int lane, const double *c_t)
{
    double p = 0.0;
    if (t.pmu) {
#pragma unroll 4
        for (int i = 0; i < d; ++i) {
            double sg = t.psig[i], th = GV(c_t, i);
            if (sg > 0.0) { double q = (th - t.pmu[i]) / sg; p = p + q * q; }
        }
    }
    return p;
}

MCX_DEV bool target_inbounds(const DevTarget &t, int d, int lane, const double *c_t)
{
    bool ok = true;
    if (t.lo || t.hi) {
#pragma unroll 4
        for (int i = 0; i < d; ++i) {
            double th = GV(c_t, i);
            if (t.lo) ok = ok && (th > t.lo[i]);
            if (t.hi) ok = ok && (th < t.hi[i]);
        }
    }
    return ok;
}

#ifndef MCX_POOLED_NB
// ... in pooled_mfma_kernel (one wave per SIMD, nothing else to issue while an attempt's chain waits: config 4 pooled 9.25e8 -> 9.65e8 at
// 8; 1: 9.04, 4: 9.21, 12: 9.55, 16: 8.99)
#define MCX_POOLED_NB 8
#endif
#ifndef MCX_POOLED_SPLIT
// pooled_mfma_kernel draws its vector in two passes (gen_normals_split): attempts first, the logarithm / root / divisions for the kept
// pairs only
#define MCX_POOLED_SPLIT 1
#endif
#ifndef MCX_POOLED_NBB
#define MCX_POOLED_NBB 4
#endif
#if MCX_POOLED_SPLIT
#define MCX_POOLED_GEN gen_normals_split<MCX_POOLED_NB, MCX_POOLED_NBB>
#else
#define MCX_POOLED_GEN gen_normals<MCX_POOLED_NB>
#endif
#ifndef MCX_RNG_NB
// polar attempts computed side by side in the kernels that wait for the generator (AM, DRAM, pooled); 4 loses at config 2 (d = 10: a vector
// is ~11 attempts)
#define MCX_RNG_NB 2
#endif
// ---------------------------------------------------------------- normals (mcmcrand.F90:60-83,166-190)
// Each lane appends accepted polar pairs to its own column of zs (global scratch, element stride 64)
// until it has d deviates; the wave loops until every participating lane is done.  The cached
// second deviate of normal_bm is honoured and left behind when d is odd.
// Returns sum(z**2) accumulated in element order (the `sum(u**2)` of MCMC_run_ram.F90:166), so the RAM
// update does not have to read the vector again.
template <int NB = 1>
MCX_DEV double gen_normals(Rng &g, double *zs_t, int lane, int d, bool participate)
{
    int k = 0;
    double su = 0.0;
    if (participate && g.saved && d > 0) { GV(zs_t, 0) = g.saved_y; su = su + g.saved_y * g.saved_y; g.saved = 0; k = 1; }
    bool need = participate && (k < d);
    if (NB == 1) {
        while (__any(need)) {
            if (need) {
                double a, b;
                if (polar_try(g, a, b)) {
                    GV(zs_t, k) = a; su = su + a * a; ++k;
                    if (k < d) { GV(zs_t, k) = b; su = su + b * b; ++k; }
                    else { g.saved_y = b; g.saved = 1; }
                }
                need = (k < d);
            }
        }
        return su;
    }
    // NB attempts per trip, side by side: the Philox blocks, the polar tests and the log / sqrt / division of NB consecutive
    // attempts of the lane's stream are independent of one another, and a kernel that waits for their dependent chains (one
    // wave per SIMD at config 2's size) gets NB chains in flight instead of one.  They are CONSUMED in order, and only as
    // many as the lane needs: an attempt past the one that completes the vector is dropped with its uniforms undrawn, so
    // the stream position, the deviates and the order of the sum are those of the one-at-a-time loop.
    while (__any(need)) {
        const uint64_t b0 = g.n >> 1;
        const bool odd = (g.n & 1) != 0;
        uint32_t w[NB + 1][4];
#pragma unroll
        for (int j = 0; j < NB; ++j) philox4x32_10((uint32_t)(b0 + j), (uint32_t)((b0 + j) >> 32), g.k0, g.k1, w[j][0], w[j][1], w[j][2],
            w[j][3]);
        if (__any(need && odd)) philox4x32_10((uint32_t)(b0 + NB), (uint32_t)((b0 + NB) >> 32), g.k0, g.k1, w[NB][0], w[NB][1], w[NB][2],
            w[NB][3]);
        else { w[NB][0] = w[NB][1] = w[NB][2] = w[NB][3] = 0u; }
        double za[NB], zb[NB];
        bool ok[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            // uniforms 2 (n/2 + j) and the next one (random_number(x), x(2): mcmcrand.F90:177)
            double x1 = odd ? bits_to_uniform(w[j][2], w[j][3]) : bits_to_uniform(w[j][0], w[j][1]);
            double x2 = odd ? bits_to_uniform(w[j + 1][0], w[j + 1][1]) : bits_to_uniform(w[j][2], w[j][3]);
            x1 = 2.0 * x1 - 1.0; x2 = 2.0 * x2 - 1.0;
            const double xx = x1 * x1 + x2 * x2;
            ok[j] = (xx < 1.0) && (xx != 0.0);
            const double z = sqrt(-2.0 * d_log(ok[j] ? xx : 0.5) / (ok[j] ? xx : 0.5));
            zb[j] = z * x1; za[j] = z * x2;
        }
        if (need) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (k < d) {
                    g.n += 2;
                    if (ok[j]) {
                        GV(zs_t, k) = za[j]; su = su + za[j] * za[j]; ++k;
                        if (k < d) { GV(zs_t, k) = zb[j]; su = su + zb[j] * zb[j]; ++k; }
                        else { g.saved_y = zb[j]; g.saved = 1; }
                    }
                }
            }
            g.cblk = 0;                                   // the half-used block (n odd) is recomputed by the next single draw
            need = (k < d);
        }
    }
    return su;
}

// The same vector in TWO passes, for a kernel whose generator is bound by instruction issue (pooled_mfma_kernel: 47 % of an iteration): a
// wave runs ~40 attempts per lane for the 25 pairs a lane of npar 50 keeps (0.785 a try, the slowest lane sets the trip count), and in the
// one-pass form every one of them pays for the logarithm, the square root and the two divisions.  Pass A makes the attempts -- the Philox
// blocks, the two uniforms, the test xx < 1 -- and parks the ACCEPTED pair's (x2, x1) where its deviates will stand; it alone moves the
// stream.  Pass B visits the parked pairs, exactly as many as the vector holds, and scales them: z = sqrt(-2 log(xx) / xx) with xx formed
// again from the same two numbers by the same two products and one sum.  Stream position, deviates, the cached second deviate and the order
// of sum(z**2): those of gen_normals.
template <int NB, int NBB = 4>
MCX_DEV double gen_normals_split(Rng &g, double *zs_t, int lane, int d, bool participate)
{
    int k = 0;
    double su = 0.0;
    if (participate && g.saved && d > 0) { GV(zs_t, 0) = g.saved_y; su = su + g.saved_y * g.saved_y; g.saved = 0; k = 1; }
    const int k0 = k;
    bool need = participate && (k < d);
    double over = 0.0;                                    // x1 of the pair whose second deviate lies past the vector's end
    // block b0 + NB -- the straddling pair's second half when the stream position is odd -- is the NEXT trip's block b0 for every lane that
    // goes on (it consumed all NB attempts): carried over instead of computed again, NB blocks per trip after the first instead of NB + 1
    uint32_t cw0 = 0u, cw1 = 0u, cw2 = 0u, cw3 = 0u;
    uint64_t cblk1 = 0;                                   // the carried block's index + 1 (0: none)
    while (__any(need)) {
        const uint64_t b0 = g.n >> 1;
        const bool odd = (g.n & 1) != 0;
        uint32_t w[NB + 1][4];
        if (__all(!need || cblk1 == b0 + 1)) { w[0][0] = cw0; w[0][1] = cw1; w[0][2] = cw2; w[0][3] = cw3; }
        else philox4x32_10((uint32_t)b0, (uint32_t)(b0 >> 32), g.k0, g.k1, w[0][0], w[0][1], w[0][2], w[0][3]);
#pragma unroll
        for (int j = 1; j <= NB; ++j) philox4x32_10((uint32_t)(b0 + j), (uint32_t)((b0 + j) >> 32), g.k0, g.k1, w[j][0], w[j][1], w[j][2],
            w[j][3]);
        cw0 = w[NB][0]; cw1 = w[NB][1]; cw2 = w[NB][2]; cw3 = w[NB][3]; cblk1 = b0 + NB + 1;
        double xa[NB], xb[NB];
        bool ok[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            double x1 = odd ? bits_to_uniform(w[j][2], w[j][3]) : bits_to_uniform(w[j][0], w[j][1]);
            double x2 = odd ? bits_to_uniform(w[j + 1][0], w[j + 1][1]) : bits_to_uniform(w[j][2], w[j][3]);
            x1 = 2.0 * x1 - 1.0; x2 = 2.0 * x2 - 1.0;
// tools/gen_bound.sh: every attempt accepted (NOT the reference's stream)
#ifdef MCX_PROBE_ALLOK
            if (!(x1 * x1 + x2 * x2 < 1.0)) { x1 *= 0.5; x2 *= 0.5; }
#endif
            const double xx = x1 * x1 + x2 * x2;
            ok[j] = (xx < 1.0) && (xx != 0.0);
            xa[j] = x2; xb[j] = x1;
        }
        if (need) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (k < d) {
                    g.n += 2;
                    if (ok[j]) {
                        GV(zs_t, k) = xa[j]; ++k;
                        if (k < d) { GV(zs_t, k) = xb[j]; ++k; }
                        else over = xb[j];
                    }
                }
            }
            g.cblk = 0;
            need = (k < d);
        }
    }
    if (participate) {
        for (int kk = k0; kk < d; kk += 2 * NBB) {
            double x1[NBB], x2[NBB], za[NBB], zb[NBB];
#pragma unroll
            for (int u = 0; u < NBB; ++u) {
                const int ka = kk + 2 * u;
                x2[u] = GV(zs_t, ka < d ? ka : d - 1);
                x1[u] = (ka + 1 < d) ? GV(zs_t, ka + 1) : over;
            }
#pragma unroll
            for (int u = 0; u < NBB; ++u) {
                const bool live = kk + 2 * u < d;
                const double xx0 = x1[u] * x1[u] + x2[u] * x2[u];
                const double xx = live ? xx0 : 0.5;
                const double z = sqrt(-2.0 * d_log(xx) / xx);
                zb[u] = z * x1[u]; za[u] = z * x2[u];
            }
#pragma unroll
            for (int u = 0; u < NBB; ++u) {
                const int ka = kk + 2 * u;
                if (ka < d) {
                    GV(zs_t, ka) = za[u]; su = su + za[u] * za[u];
                    if (ka + 1 < d) { GV(zs_t, ka + 1) = zb[u]; su = su + zb[u] * zb[u]; }
                    else { g.saved_y = zb[u]; g.saved = 1; }
                }
            }
        }
    }
    return su;
}

} // namespace mcx
