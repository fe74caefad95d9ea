// mcx_scam.hpp -- MCMC_run_scam (MCMC_run_scam.F90:38-138): per-chain rotations (scam_kernel, scam_mw_kernel) and the pooled rotation on
// the f64 matrix cores (scam_pooled_kernel, scam_pooled12_kernel); the lane state the phase kernels share (LaneState)
// (one of the family headers mcx_kernels.hpp includes, in this order: mcx_common, mcx_products, mcx_step, mcx_scam, mcx_pooled, mcx_phase,
// mcx_adapt, mcx_svd, mcx_moments)
#pragma once
#include "mcx_step.hpp"

namespace mcx {

// ---------------------------------------------------------------- host-callback targets
// When ssfunction / priorfun / checkbounds are host functions of the user (external_inc.h:4-33) one
// iteration is cut where the reference calls them (MCMC_run.F90:47,55-56,69,74-75): phase 0 proposes,
// the host evaluates the candidates of all chains in chain order, phase 1 decides (and proposes the DR
// try), the host evaluates again, phase 2 decides the DR try and finishes the iteration.  Same device
// functions as step_kernel; per-lane state round-trips through HBM between phases.
enum { HX_SS2 = 0, HX_PRI2, HX_REJECT, HX_STAGE2, HX_DRMOVED, HX_SU, HX_CRIT, HX_MOVED, NHX };
enum { HE_INB = 0, HE_PRI, HE_SS, NHE };

struct LaneState {
    Rng g;
    double ss1, pri1, sigma2, alpha12;
    uint32_t stayed, bnd, chainind, curcount, status, dracc, drtries, pdesc;
};
MCX_DEV void lane_load(const EngineDev &E, int tile, int lane, LaneState &L)
{
    L.g.k0 = E.k0; L.g.k1 = E.chain_id0 + (uint32_t)(tile * 64 + lane);
    L.g.n = TIDX(E.rngn, tile, 1, 0, lane); L.g.cblk = 0; L.g.c2 = 0; L.g.c3 = 0;
    L.g.saved = (int)TIDX(E.ictr, tile, NICTR, I_SAVED, lane);
    L.g.saved_y = TIDX(E.scal, tile, NSCAL, S_SAVEDY, lane);
    L.ss1 = TIDX(E.scal, tile, NSCAL, S_SS1, lane); L.pri1 = TIDX(E.scal, tile, NSCAL, S_PRI1, lane);
    L.sigma2 = TIDX(E.scal, tile, NSCAL, S_SIGMA2, lane); L.alpha12 = TIDX(E.scal, tile, NSCAL, S_ALPHA12, lane);
    L.stayed = TIDX(E.ictr, tile, NICTR, I_STAYED, lane); L.bnd = TIDX(E.ictr, tile, NICTR, I_BNDSTAYED, lane);
    L.chainind = TIDX(E.ictr, tile, NICTR, I_CHAININD, lane); L.curcount = TIDX(E.ictr, tile, NICTR, I_CURCOUNT, lane);
    L.status = TIDX(E.ictr, tile, NICTR, I_STATUS, lane);
    L.dracc = TIDX(E.ictr, tile, NICTR, I_DRACC, lane); L.drtries = TIDX(E.ictr, tile, NICTR, I_DRTRIES, lane);
    L.pdesc = TIDX(E.ictr, tile, NICTR, I_PDESC, lane);
}
MCX_DEV void lane_store(const EngineDev &E, int tile, int lane, const LaneState &L)
{
    TIDX(E.rngn, tile, 1, 0, lane) = L.g.n;
    TIDX(E.ictr, tile, NICTR, I_SAVED, lane) = (uint32_t)L.g.saved;
    TIDX(E.scal, tile, NSCAL, S_SAVEDY, lane) = L.g.saved_y;
    TIDX(E.scal, tile, NSCAL, S_SS1, lane) = L.ss1; TIDX(E.scal, tile, NSCAL, S_PRI1, lane) = L.pri1;
    TIDX(E.scal, tile, NSCAL, S_SIGMA2, lane) = L.sigma2; TIDX(E.scal, tile, NSCAL, S_ALPHA12, lane) = L.alpha12;
    TIDX(E.ictr, tile, NICTR, I_STAYED, lane) = L.stayed; TIDX(E.ictr, tile, NICTR, I_BNDSTAYED, lane) = L.bnd;
    TIDX(E.ictr, tile, NICTR, I_CHAININD, lane) = L.chainind; TIDX(E.ictr, tile, NICTR, I_CURCOUNT, lane) = L.curcount;
    TIDX(E.ictr, tile, NICTR, I_STATUS, lane) = L.status;
    TIDX(E.ictr, tile, NICTR, I_DRACC, lane) = L.dracc; TIDX(E.ictr, tile, NICTR, I_DRTRIES, lane) = L.drtries;
    TIDX(E.ictr, tile, NICTR, I_PDESC, lane) = L.pdesc;
}

// scam_fast: newpar_k = oldpar_k + delta U(k,j), elements k0, k0 + kstep, ... (one fma each; U column-major per chain)
MCX_DEV void scam_fast_propose(const double *Ut, const double *theta_t, double *cand_t, int lane, int d, int j, double delta, int p0 = 0,
    int pstep = 1)
{
    const double *col = Ut + (size_t)j * d * 64;
    for (int K0 = p0 * PW; K0 < d; K0 += pstep * PW) {
        double u[PW], th[PW];
#pragma unroll
        for (int q = 0; q < PW; ++q) { const int k = K0 + q < d ? K0 + q : d - 1; u[q] = LDNT(col, k); th[q] = GV(theta_t, k); }
#pragma unroll
        for (int q = 0; q < PW; ++q) if (K0 + q < d) GV(cand_t, K0 + q) = dfma(delta, u[q], th[q]);
    }
}

// ---------------------------------------------------------------- MCMC_run_scam (MCMC_run_scam.F90:38-88)
// One outer iteration = d componentwise Metropolis sub-steps in the rotated basis: rot = U'theta (dgemv 'T'),
// rot_j += N(0,1) std_j, theta' = U rot (dgemv 'N'), full ss evaluation, alpha, reject (MCMC_propose_sc :94-117).
// One chain row per outer iteration.  U (full d x d per chain) is streamed twice per sub-step.
#ifndef MCX_SCAM_WAVES
#define MCX_SCAM_WAVES 2
#endif
__global__ __launch_bounds__(64, MCX_SCAM_WAVES) void scam_kernel(EngineDev E, int it0, int it1,
                                                     const double *__restrict__ g_mu, const double *__restrict__ g_lamT)
{
    const int lane = threadIdx.x, tile = blockIdx.x, d = E.d;
    double *theta_t = E.theta + (size_t)tile * d * 64;
    double *cand_t = E.cand + (size_t)tile * d * 64;
    double *rot_t = E.cs + (size_t)tile * 2 * d * 64;
    const double *Ut = E.Rf + (size_t)tile * d * d * 64;
    const double *std_t = E.qstd + (size_t)tile * d * 64;
    LaneState L;
    lane_load(E, tile, lane, L);
    for (int it = it0; it <= it1; ++it) {
        bool rejall = true;
        for (int j = 0; j < d; ++j) {
            if (E.scam_fast) {
                const double zj = rng_normal(L.g) * GV(std_t, j);
                scam_fast_propose(Ut, theta_t, cand_t, lane, d, j, zj);
            } else {
                gemvT_panels<true>(Ut, theta_t, rot_t, lane, d);
                const double zj = rng_normal(L.g) * GV(std_t, j);
                GV(rot_t, j) = GV(rot_t, j) + zj;
                gemvN_panels<true>(Ut, rot_t, cand_t, nullptr, lane, d, true);
            }
            bool inb = target_inbounds(E.tgt, d, lane, cand_t);
            double pri2 = target_prior(E.tgt, d, lane, cand_t);
            double ss2 = target_ss<false>(E.tgt, d, lane, cand_t, g_mu, g_lamT);
            bool reject;
            if (!inb) { L.bnd += 1; L.alpha12 = 0.0; reject = true; }
            else {
                L.alpha12 = d_alpha(L.ss1, L.pri1, ss2, pri2, L.sigma2);
                reject = true;
                if (L.alpha12 >= 1.0) reject = false;
                else if (L.alpha12 > 0.0) { double u = rng_uniform(L.g); if (u <= L.alpha12) reject = false; }
            }
            if (!reject) {
                L.ss1 = ss2; L.pri1 = pri2; rejall = false;
                copy_vec(theta_t, cand_t, nullptr, lane, d);
            }
        }
        if (rejall) { L.stayed += 1; L.curcount += 1; }
        else { L.chainind += 1; L.curcount = 1; }
        if (E.updatesigma) {
            double gm = rng_gamma(L.g, E.gam_shape, 2.0 / (E.N0S02 + L.ss1));
            L.sigma2 = 1.0 / gm;
        }
        unsigned long long ballot = __ballot(!rejall);
        const int slot = it % E.wcap;
        if (E.hist) {
            if (!rejall) {
                double *h = E.hist + ((size_t)tile * E.wcap + slot) * (size_t)E.hs * 64;
                for (int k = 0; k < d; ++k) GV(h, k) = GV(theta_t, k);
                GV(h, d) = L.ss1;
            }
            if (lane == 0) E.wacc[(size_t)tile * E.wcap + slot] = ballot;
            if (E.record_s2) E.s2hist[((size_t)tile * E.wcap + slot) * 64 + lane] = L.sigma2;
        }
        if (E.accmask && lane == 0) E.accmask[(size_t)(it - 1) * E.ntiles + tile] = ballot;
    }
    lane_store(E, tile, lane, L);
}

// The same iteration with NW waves per tile (a workgroup of 64 NW threads; lane = chain, wave = a share of the work): for
// chain counts that leave most of the chip idle at one wave per tile (the reference's own use is ONE chain), where a
// sub-step is bound by the latency of one wave's loads -- 2 d^2 x 512 bytes streamed with ~16 kB in flight.  Every output
// element of the two products is its own fma chain (gemvT: one per column, gemvN: one per row), and the Gaussian target's
// blocks of 16 rows are independent up to the running sum of their q_k, so the waves share panels / blocks without
// changing one operation; wave 0 owns the per-chain scalar state (stream, ss1, counters), draws, decides, and hands the
// deviate and the accept flag to the others through LDS.  Vectors stay in the per-chain global scratch (the workgroup's
// waves run on one CU and meet at workgroup barriers).  lds: [16 nblk][64] partial chains, [64] deviates, [64] flags.
template <int NW>
__global__ __launch_bounds__(64 * NW) void scam_mw_kernel(EngineDev E, int it0, int it1,
                                                          const double *__restrict__ g_mu, const double *__restrict__ g_lamT)
{
    extern __shared__ double lds_mw[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, tile = blockIdx.x, d = E.d;
    const int nblk = (d + 15) / 16;
    double *Q = lds_mw, *zl = lds_mw + (size_t)4 * nblk * 64, *fl = zl + 64;
    double *theta_t = E.theta + (size_t)tile * d * 64;
    double *cand_t = E.cand + (size_t)tile * d * 64;
    double *rot_t = E.cs + (size_t)tile * 2 * d * 64;
    const double *Ut = E.Rf + (size_t)tile * d * d * 64;
    const double *std_t = E.qstd + (size_t)tile * d * 64;
    const bool gauss = (E.tgt.kind == TGT_GAUSS);
    LaneState L;
    if (w == 0) lane_load(E, tile, lane, L);
    for (int it = it0; it <= it1; ++it) {
        bool rejall = true;
        for (int j = 0; j < d; ++j) {
            if (w == 0) zl[lane] = rng_normal(L.g) * GV(std_t, j);           // the sub-step's first draw (MCMC_run_scam.F90:108)
            if (E.scam_fast) {
                __syncthreads();
                scam_fast_propose(Ut, theta_t, cand_t, lane, d, j, zl[lane], w, NW);
            } else {
                // many waves: fewer rows in flight each (registers)
                gemvT_panels<true, (NW >= 8 ? 2 : 0)>(Ut, theta_t, rot_t, lane, d, w, NW);
                __syncthreads();
                if (w == (j / PW) % NW) GV(rot_t, j) = GV(rot_t, j) + zl[lane];  // by the wave that wrote rot_j
                __syncthreads();
                gemvN_panels<true, (NW >= 8 ? 2 : 0)>(Ut, rot_t, cand_t, nullptr, lane, d, true, w, NW);
            }
            __syncthreads();
            if (gauss) {
                for (int b = w; b < nblk; b += NW) {
                    double q[4];
                    gauss_block_q(d, lane, cand_t, g_mu, g_lamT, 16 * b, q);
#pragma unroll
                    for (int k = 0; k < 4; ++k) Q[(size_t)(4 * b + k) * 64 + lane] = q[k];
                }
                __syncthreads();
            }
            if (w == 0) {
                bool inb = target_inbounds(E.tgt, d, lane, cand_t);
                double pri2 = target_prior(E.tgt, d, lane, cand_t);
                double ss2 = 0.0;
                if (gauss) { for (int e = 0; e < 4 * nblk;
                    ++e) if (16 * (e >> 2) + (e & 3) < d) ss2 = (e == 0) ? Q[lane] : ss2 + Q[(size_t)e * 64 + lane]; }
                else ss2 = target_ss<false>(E.tgt, d, lane, cand_t, g_mu, g_lamT);
                bool reject;
                if (!inb) { L.bnd += 1; L.alpha12 = 0.0; reject = true; }
                else {
                    L.alpha12 = d_alpha(L.ss1, L.pri1, ss2, pri2, L.sigma2);
                    reject = true;
                    if (L.alpha12 >= 1.0) reject = false;
                    else if (L.alpha12 > 0.0) { double u = rng_uniform(L.g); if (u <= L.alpha12) reject = false; }
                }
                if (!reject) { L.ss1 = ss2; L.pri1 = pri2; rejall = false; }
                fl[lane] = reject ? 0.0 : 1.0;
            }
            __syncthreads();
            if (fl[lane] != 0.0) for (int K0 = w * PW; K0 < d; K0 += NW * PW) {
                double v[PW];
#pragma unroll
                for (int u = 0; u < PW; ++u) v[u] = GV(cand_t, K0 + (K0 + u < d ? u : 0));
#pragma unroll
                for (int u = 0; u < PW; ++u) if (K0 + u < d) GV(theta_t, K0 + u) = v[u];
            }
            __syncthreads();
        }
        if (w == 0) {
            if (rejall) { L.stayed += 1; L.curcount += 1; }
            else { L.chainind += 1; L.curcount = 1; }
            if (E.updatesigma) {
                double gm = rng_gamma(L.g, E.gam_shape, 2.0 / (E.N0S02 + L.ss1));
                L.sigma2 = 1.0 / gm;
            }
            unsigned long long ballot = __ballot(!rejall);
            const int slot = it % E.wcap;
            if (E.hist) {
                if (!rejall) {
                    double *h = E.hist + ((size_t)tile * E.wcap + slot) * (size_t)E.hs * 64;
                    copy_vec(h, theta_t, nullptr, lane, d);
                    GV(h, d) = L.ss1;
                }
                if (lane == 0) E.wacc[(size_t)tile * E.wcap + slot] = ballot;
                if (E.record_s2) E.s2hist[((size_t)tile * E.wcap + slot) * 64 + lane] = L.sigma2;
            }
            if (E.accmask && lane == 0) E.accmask[(size_t)(it - 1) * E.ntiles + tile] = ballot;
        }
    }
    if (w == 0) lane_store(E, tile, lane, L);
}

// ---------------------------------------------------------------- pooled SCAM: one rotation shared by all chains
// MCMC_run_scam with ONE rotation U (and one qcovstd) for every chain of the node (pooled mode), on the matrix cores.
//
// out(o, c) = sum_s M[s*d + o] X(s, c) for every output row o and chain c of a tile, s ascending, one fma chain per
// (o, c) -- the order of gemvT_panels / gemvN_panels / the Gaussian y rows.  v_mfma_f64_16x16x4_f64 accumulates its
// four products as an ascending fma chain (checked bit for bit on gfx950, tools/mfma_f64_probe.hip), so D = A B + D
// repeated over blocks of four s IS that chain.  A = M' (16 outputs x 4 s, from the shared table, L2-resident),
// B = X (4 s x 16 chains, from the workgroup's LDS vector; d4 = 4*ceil(d/4) rows, the pad rows zero; M has d4 rows,
// pad rows zero, and PWS doubles of slack).
//
// A workgroup is nw waves that share one tile of 64 chains.  Every wave owns four 16x16 (output block x chain group)
// result tiles -- "slots" -- and keeps them in registers in the MFMA C layout (row = 16*block + (lane>>4) + 4r,
// chain = 16*group + (lane&15)): a block wave (w < ntw) owns output block w for all four chain groups (one A, four
// B per k-block), each of the last four waves owns chain group w-ntw of the leftover blocks ntw.. (one B, up to four
// A) -- so every SIMD (wave mod 4) runs the same number of MFMAs.  The products of a sub-step chain through LDS only:
// theta -> X -> rot (registers) -> X -> theta' (registers, kept for the accept) -> X = theta'-mu -> y (registers) ->
// per-lane partial chains q of ss (mcxt_ss_gauss's order is exactly this layout) -> LDS; the last wave carries the
// per-chain scalar state, sums the q, does prior / bounds / alpha / accept and hands the normal deviate and the accept
// flag of each chain to the others through LDS.  Arithmetic per chain is operation for operation that of scam_kernel.
constexpr int PWS = 16;
typedef double mcx_d4 __attribute__((ext_vector_type(4)));
typedef double mcx_d2 __attribute__((ext_vector_type(2)));

// BW: block wave (blk0 = its block, slot = chain group; XS: a fifth slot, group xgrp of block xblk); else group wave (slot s = block
// blk0+s, s < NS)
template <bool BW, int NS, bool XS = false>
MCX_DEV void mfma_slots(const double *__restrict__ M, const double *X, int lane, int d, int d4, int blk0, int grp, mcx_d4 (&c)[XS ? 5 : 4],
                        int xblk = 0, int xgrp = 0)
{
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int s = 0; s < (XS ? 5 : 4); ++s) c[s] = mcx_d4{0.0, 0.0, 0.0, 0.0};
    const double *__restrict__ ap = M + (size_t)lk * d + 16 * blk0 + li;
    const double *xp = X + lk * 64 + li + (BW ? 0 : 16 * grp);
    int s0 = 0;
    if (BW) {
        const double *__restrict__ axp = M + (size_t)lk * d + 16 * xblk + li;      // XS: the fifth slot's A operand
        for (; s0 + 16 <= d4; s0 += 16) {               // four k-blocks per trip: the four (eight) A loads go out together
            double a[4], ax[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = ap[(size_t)(s0 + 4 * u) * d]; if (XS) ax[u] = axp[(size_t)(s0 + 4 * u) * d]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double *xq = xp + (s0 + 4 * u) * 64;
#pragma unroll
                for (int g = 0; g < 4; ++g) c[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], xq[16 * g], c[g], 0, 0, 0);
                if (XS) c[XS ? 4 : 0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[u], xq[16 * xgrp], c[XS ? 4 : 0], 0, 0, 0);
            }
        }
        for (; s0 < d4; s0 += 4) {
            const double a = ap[(size_t)s0 * d];
            const double *xq = xp + s0 * 64;
#pragma unroll
            for (int g = 0; g < 4; ++g) c[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xq[16 * g], c[g], 0, 0, 0);
            if (XS) c[XS ? 4 : 0] = __builtin_amdgcn_mfma_f64_16x16x4f64(axp[(size_t)s0 * d], xq[16 * xgrp], c[XS ? 4 : 0], 0, 0, 0);
        }
    } else {
        // KU k-blocks per trip, up to four A each: a trip waits for its loads once, and a wave with one or two slots has few
        // MFMAs to put behind them -- with two k-blocks per trip the four group waves were the last at every barrier
        // (64 us per sub-step against the block waves' 54 at d = 200); eight A loads in flight per trip whatever NS is
        constexpr int KU = NS <= 1 ? 8 : (NS == 2 ? 4 : 2);
        for (; s0 + 4 * KU <= d4; s0 += 4 * KU) {
            double a[KU][NS > 0 ? NS : 1], bq[KU];
#pragma unroll
            for (int u = 0; u < KU; ++u) {
#pragma unroll
                for (int s = 0; s < NS; ++s) a[u][s] = ap[(size_t)(s0 + 4 * u) * d + 16 * s];
                bq[u] = xp[(s0 + 4 * u) * 64];
            }
#pragma unroll
            for (int u = 0; u < KU; ++u)
#pragma unroll
                for (int s = 0; s < NS; ++s) c[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][s], bq[u], c[s], 0, 0, 0);
        }
        for (; s0 < d4; s0 += 4) {
            const double bq = xp[s0 * 64];
#pragma unroll
            for (int s = 0; s < NS; ++s) c[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[(size_t)s0 * d + 16 * s], bq, c[s], 0, 0, 0);
        }
    }
}
// Element (slot s, register r) of a lane: output row o = 16*block + (lane>>4) + 4r, chain c = 16*group + (lane&15); its
// offset o*64 + c in a tile-interleaved vector (and in X) is e0 + (BW ? 16 s : 1024 s) + 256 r.  X has 16*nt rows, so
// every element has an LDS home; rows >= d are written as zeros (the k loop reads the rows < d4 only).
// SC: the scalar wave (the last one).  A template parameter, so that the other fifteen waves carry
template <bool BW, int NS, bool SC, bool XS = false>
                                      // neither the generator nor the per-chain state: at 128 registers a wave they spilled around their
                                      // MFMAs XS (scam_pooled12_kernel): a block wave with a FIFTH slot, chain group xgrp of block xblk
MCX_DEV void scam_pooled_body(const EngineDev &E, int it0, int it1, double *X, int lane, int w, int nw, int blk0, int grp,
                              const double *__restrict__ g_mu, const double *__restrict__ g_lamT,
                              const double *__restrict__ g_U, const double *__restrict__ g_UT, const double *__restrict__ g_std,
                              int xblk = 0, int xgrp = 0)
{
    const int tile = blockIdx.x, d = E.d, d4 = (d + 3) & ~3, nt = (d + 15) >> 4, li = lane & 15, lk = lane >> 4;
    double *Q = X + (size_t)nt * 16 * 64;                                       // [4*nt][64] partial ss chains
    double *zb = Q + (size_t)nt * 4 * 64, *fl = zb + 64;                      // per chain: the deviate, the accept flag
    double *mul = fl + 64;                                                      // the target's mean, [16 nt]: read at every third fill
    constexpr bool sc = SC;                                                     // the scalar wave
    const bool gauss = (E.tgt.kind == TGT_GAUSS);
    const bool cand_global = !gauss || E.tgt.pmu || E.tgt.lo || E.tgt.hi;       // prior / bounds / other targets read theta' per chain
    static_assert(!XS || (BW && NS == 4), "the fifth slot belongs to a block wave");
    constexpr int nsl = XS ? 5 : NS, NA = XS ? 5 : 4;
    const int e0 = (16 * blk0 + lk) * 64 + (BW ? 0 : 16 * grp) + li, o0 = 16 * blk0 + lk, c0 = (BW ? 0 : 16 * grp) + li;
    const int ex = (16 * xblk + lk) * 64 + 16 * xgrp + li, ox = 16 * xblk + lk, cx = 16 * xgrp + li;      // the fifth slot
#define EOFF(s, r) ((XS && (s) == 4) ? ex + 256 * (r) : e0 + (BW ? 16 : 1024) * (s) + 256 * (r))
#define EROW(s, r) ((XS && (s) == 4) ? ox + 4 * (r) : o0 + (BW ? 0 : 16) * (s) + 4 * (r))
#define ECH(s) ((XS && (s) == 4) ? cx : c0 + (BW ? 16 : 0) * (s))
    double *theta_t = E.theta + (size_t)tile * d * 64;
    double *cand_t = E.cand + (size_t)tile * d * 64;
    LaneState L;
    if (sc) lane_load(E, tile, lane, L);
    // The group waves own one or two tile sets whose MFMAs form ONE dependent chain per product; at equal priority the (older)
    // block waves' sixteen independent MFMAs per trip win the matrix pipe and the chain only runs once they are done -- the
    // whole workgroup then waits ~2 us per product at the barrier.  Raised priority lets the chain interleave.
    if (!BW) __builtin_amdgcn_s_setprio(2);
    if (gauss) { for (int o = w * 64 + lane; o < 16 * nt; o += nw * 64) mul[o] = o < d ? g_mu[o] : 0.0; }
    mcx_d4 cand[NA], cc[NA], th[NA];
    // The chains' state: every lane keeps the elements of its slots in registers across the sub-steps (they are the ones it
    // fills into X and the ones it replaces on an accept) and writes them back once per iteration.
#pragma unroll
    for (int s = 0; s < NA; ++s)
        if (s < nsl) {
#pragma unroll
            for (int r = 0; r < 4; ++r) th[s][r] = theta_t[EROW(s, r) < d ? EOFF(s, r) : e0];
        }
#ifdef MCX_PHASE_PROF
    unsigned long long ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tq = wall_clock64();
#define PH(i) { unsigned long long tn = wall_clock64(); ph[i] += tn - tq; tq = tn; }
#else
#define PH(i)
#endif
    for (int it = it0; it <= it1; ++it) {
        bool rejall = true;
        for (int j = 0; j < d; ++j) {
          if (E.scam_fast) {
            // opt-in (mcmcx_config::scam_fast): theta' = theta + delta U(:,j) from the registers -- no rotation products at all
            // (g_U == nullptr: per-chain rotations -- the column comes from the chain's own factor, the target still runs on the
            //  matrix cores: what the lane-per-chain kernels cannot do for it, they re-read the candidate once per 8 rows)
            const bool pc = (g_U == nullptr);
            if (sc) zb[lane] = rng_normal(L.g) * (pc ? TIDX(E.qstd, tile, d, j, lane) : g_std[j]);
            __syncthreads();
#pragma unroll
            for (int s = 0; s < NA; ++s) {
                if (s < nsl) {
                    const double zj = zb[ECH(s)];
                    double uc[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int o = EROW(s, r) < d ? EROW(s, r) : 0;
                        uc[r] = pc ? __builtin_nontemporal_load(&E.Rf[((size_t)tile * d * d + (size_t)j * d + o) * 64 + ECH(s)])
                            : g_U[(size_t)j * d + o];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) cand[s][r] = EROW(s, r) < d ? dfma(zj, uc[r], th[s][r]) : 0.0;
                }
            }
          } else {
#pragma unroll
            for (int s = 0; s < NA; ++s) {                                       // X = theta
                if (s < nsl) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) X[EOFF(s, r)] = EROW(s, r) < d ? th[s][r] : 0.0;
                }
            }
            PH(0)
            __syncthreads();
            PH(7)
            // the sub-step's deviate is not needed before the second fill: the scalar wave draws it while the first product runs
            // (its own share of the product is one tile set) instead of holding everybody at the barrier above
            if (sc) zb[lane] = rng_normal(L.g) * g_std[j];
            mfma_slots<BW, NS, XS>(g_UT, X, lane, d, d4, blk0, grp, cc, xblk, xgrp);           // rot = U'theta
            PH(2)
            __syncthreads();
            PH(8)
#pragma unroll
            for (int s = 0; s < NA; ++s) {
                if (s < nsl) {
                    const double zj = zb[ECH(s)];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int o = EROW(s, r);
                        double v = cc[s][r];
                        if (o == j) v = v + zj;
                        X[EOFF(s, r)] = o < d ? v : 0.0;
                    }
                }
            }
            PH(3)
            __syncthreads();
            PH(9)
            mfma_slots<BW, NS, XS>(g_U, X, lane, d, d4, blk0, grp, cand, xblk, xgrp);          // theta' = U rot
            PH(2)
          }
            if (cand_global) {
#pragma unroll
                for (int s = 0; s < NA; ++s)
                    if (s < nsl) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (EROW(s, r) < d) cand_t[EOFF(s, r)] = cand[s][r];
                    }
            }
            __syncthreads();
            if (gauss) {
#pragma unroll
                for (int s = 0; s < NA; ++s)
                    if (s < nsl) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const int o = EROW(s, r); X[EOFF(s, r)] = o < d ? cand[s][r] - mul[o] : 0.0; }
                    }
                PH(3)
                __syncthreads();
                PH(10)
                mfma_slots<BW, NS, XS>(g_lamT, X, lane, d, d4, blk0, grp, cc, xblk, xgrp);     // y = Lam v
                PH(2)
#pragma unroll
                for (int s = 0; s < NA; ++s) {                                   // q_(block, lane>>4) = chain over r of y v
                    if (s < nsl) {
                        double q = cc[s][0] * X[EOFF(s, 0)];
#pragma unroll
                        for (int r = 1; r < 4; ++r) { const double t = dfma(cc[s][r], X[EOFF(s, r)], q); q = EROW(s, r) < d ? t : q; }
                        if (EROW(s, 0) < d) Q[(size_t)(EROW(s, 0) >> 4) * 256 + (EROW(s, 0) & 3) * 64 + ECH(s)] = q;
                    }
                }
                PH(4)
                __syncthreads();
                PH(11)
            }
            if (sc) {
                bool inb = true; double pri2 = 0.0, ss2 = 0.0;
                if (cand_global) { inb = target_inbounds(E.tgt, d, lane, cand_t); pri2 = target_prior(E.tgt, d, lane, cand_t); }
                if (gauss) {
                    ss2 = Q[lane];
#pragma unroll 4
                    for (int e = 1; e < 4 * nt; ++e) if (16 * (e >> 2) + (e & 3) < d) ss2 = ss2 + Q[(size_t)e * 64 + lane];
                } else ss2 = target_ss<false>(E.tgt, d, lane, cand_t, g_mu, g_lamT);
                bool reject;
                if (!inb) { L.bnd += 1; L.alpha12 = 0.0; reject = true; }
                else {
                    L.alpha12 = d_alpha(L.ss1, L.pri1, ss2, pri2, L.sigma2);
                    reject = true;
                    if (L.alpha12 >= 1.0) reject = false;
                    else if (L.alpha12 > 0.0) { double u = rng_uniform(L.g); if (u <= L.alpha12) reject = false; }
                }
                if (!reject) { L.ss1 = ss2; L.pri1 = pri2; rejall = false; }
                fl[lane] = reject ? 0.0 : 1.0;
            }
            PH(5)
            __syncthreads();
            PH(12)
            // accepted chains: theta = theta' (each lane its own elements; the next sub-step reloads exactly those)
#pragma unroll
            for (int s = 0; s < NA; ++s)
                if (s < nsl) {
                    const bool acc = fl[ECH(s)] != 0.0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) th[s][r] = acc ? cand[s][r] : th[s][r];
                }
        }
#pragma unroll
        for (int s = 0; s < NA; ++s)                                             // the iteration's state: for the history row below, the
            if (s < nsl) {                                                      // pooled moments and the next launch
#pragma unroll
                for (int r = 0; r < 4; ++r) if (EROW(s, r) < d) theta_t[EOFF(s, r)] = th[s][r];
            }
        __syncthreads();
        if (sc) {
            if (rejall) { L.stayed += 1; L.curcount += 1; }
            else { L.chainind += 1; L.curcount = 1; }
            if (E.updatesigma) {
                double gm = rng_gamma(L.g, E.gam_shape, 2.0 / (E.N0S02 + L.ss1));
                L.sigma2 = 1.0 / gm;
            }
            unsigned long long ballot = __ballot(!rejall);
            const int slot = it % E.wcap;
            if (E.hist) {
                if (!rejall) {
                    double *h = E.hist + ((size_t)tile * E.wcap + slot) * (size_t)E.hs * 64;
                    for (int k = 0; k < d; ++k) GV(h, k) = GV(theta_t, k);
                    GV(h, d) = L.ss1;
                }
                if (lane == 0) E.wacc[(size_t)tile * E.wcap + slot] = ballot;
                if (E.record_s2) E.s2hist[((size_t)tile * E.wcap + slot) * 64 + lane] = L.sigma2;
            }
            if (E.accmask && lane == 0) E.accmask[(size_t)(it - 1) * E.ntiles + tile] = ballot;
        }
    }
    if (sc) lane_store(E, tile, lane, L);
#ifdef MCX_PHASE_PROF
    PH(6)
    if (tile == 0 && lane == 0 && (w == 0
        || sc)) printf("wave %d x10ns: theta+fill %llu mfma %llu fills %llu q %llu scalar %llu accept %llu | barrier waits after: fill0 %llu P1 %llu fill1 %llu P2 %llu fill2 %llu P3q %llu scalar %llu\n", w, ph[0], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7], ph[8], ph[9], ph[10], ph[11], ph[12], ph[13]);
#endif
#undef PH
#undef EOFF
#undef EROW
#undef ECH
}

__global__ __launch_bounds__(1024, 1) void scam_pooled_kernel(EngineDev E, int it0, int it1,
                                                             const double *__restrict__ g_mu, const double *__restrict__ g_lamT,
                                                             const double *__restrict__ g_U, const double *__restrict__ g_UT,
                                                             const double *__restrict__ g_std)
{
    extern __shared__ double X[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = (int)(blockDim.x >> 6);
    const int nt = (E.d + 15) >> 4, ntw = nw - 4;                               // ntw block waves own blocks 0..ntw-1
    if (w < ntw) scam_pooled_body<true, 4, false>(E, it0, it1, X, lane, w, nw, w, 0, g_mu, g_lamT, g_U, g_UT, g_std);
    else if (w == nw - 1) switch (nt - ntw) {                                   // the scalar wave
        case 0: scam_pooled_body<false, 0, true>(E, it0, it1, X, lane, w, nw, ntw, w - ntw, g_mu, g_lamT, g_U, g_UT, g_std); break;
        case 1: scam_pooled_body<false, 1, true>(E, it0, it1, X, lane, w, nw, ntw, w - ntw, g_mu, g_lamT, g_U, g_UT, g_std); break;
        case 2: scam_pooled_body<false, 2, true>(E, it0, it1, X, lane, w, nw, ntw, w - ntw, g_mu, g_lamT, g_U, g_UT, g_std); break;
        case 3: scam_pooled_body<false, 3, true>(E, it0, it1, X, lane, w, nw, ntw, w - ntw, g_mu, g_lamT, g_U, g_UT, g_std); break;
        default: scam_pooled_body<false, 4, true>(E, it0, it1, X, lane, w, nw, ntw, w - ntw, g_mu, g_lamT, g_U, g_UT, g_std); break;
    }
    else switch (nt - ntw) {
        case 0: scam_pooled_body<false, 0, false>(E, it0, it1, X, lane, w, nw, ntw, w - ntw, g_mu, g_lamT, g_U, g_UT, g_std); break;
        case 1: scam_pooled_body<false, 1, false>(E, it0, it1, X, lane, w, nw, ntw, w - ntw, g_mu, g_lamT, g_U, g_UT, g_std); break;
        case 2: scam_pooled_body<false, 2, false>(E, it0, it1, X, lane, w, nw, ntw, w - ntw, g_mu, g_lamT, g_U, g_UT, g_std); break;
        case 3: scam_pooled_body<false, 3, false>(E, it0, it1, X, lane, w, nw, ntw, w - ntw, g_mu, g_lamT, g_U, g_UT, g_std); break;
        default: scam_pooled_body<false, 4, false>(E, it0, it1, X, lane, w, nw, ntw, w - ntw, g_mu, g_lamT, g_U, g_UT, g_std); break;
    }
}

// The same sub-step with TWELVE waves for 13..15 output blocks (npar 193..240): every wave is a block wave (blocks 0..11, four chain
// groups each), and the slots of the blocks 12.. -- four per block, one chain group each -- ride as a FIFTH slot on the waves 0, 1, 2, ...
// (wave w: group w % 4 of block 12 + w / 4), so every SIMD (wave mod 4) still runs nt tile sets per product.  Three waves per SIMD have
// 170 registers each instead of 128: the three 32-register tile sets of a block wave (state, candidate, product) no longer spill around
// the products.  The last wave carries the per-chain scalar state on top of its block.
__global__ __launch_bounds__(768, 1) void scam_pooled12_kernel(EngineDev E, int it0, int it1,
                                                              const double *__restrict__ g_mu, const double *__restrict__ g_lamT,
                                                              const double *__restrict__ g_U, const double *__restrict__ g_UT,
                                                              const double *__restrict__ g_std)
{
    extern __shared__ double X[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = 12;
    const int nt = (E.d + 15) >> 4, nx = 4 * (nt - 12);                          // nx extra slots, on the waves 0 .. nx - 1
    const bool xs = w < nx;
    const int xblk = 12 + (w >> 2), xgrp = w & 3;
    if (w == nw - 1) {
        if (xs) scam_pooled_body<true, 4, true, true>(E, it0, it1, X, lane, w, nw, w, 0, g_mu, g_lamT, g_U, g_UT, g_std, xblk, xgrp);
        else scam_pooled_body<true, 4, true, false>(E, it0, it1, X, lane, w, nw, w, 0, g_mu, g_lamT, g_U, g_UT, g_std);
    } else {
        if (xs) scam_pooled_body<true, 4, false, true>(E, it0, it1, X, lane, w, nw, w, 0, g_mu, g_lamT, g_U, g_UT, g_std, xblk, xgrp);
        else scam_pooled_body<true, 4, false, false>(E, it0, it1, X, lane, w, nw, w, 0, g_mu, g_lamT, g_U, g_UT, g_std);
    }
}

} // namespace mcx
