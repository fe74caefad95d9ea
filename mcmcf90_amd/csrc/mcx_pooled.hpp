// mcx_pooled.hpp -- pooled AM / RAM / ER / DR on the f64 matrix cores (pooled_mfma_kernel): one wave per tile, the shared tables' products
// as v_mfma_f64_16x16x4_f64 tiles (one of the family headers mcx_kernels.hpp includes, in this order: mcx_common, mcx_products, mcx_step,
// mcx_scam, mcx_pooled, mcx_phase, mcx_adapt, mcx_svd, mcx_moments)
#pragma once
#include "mcx_scam.hpp"

namespace mcx {

// ---------------------------------------------------------------- pooled AM on the matrix cores
// One wave = one tile of 64 chains, lane = chain for everything sequential (random numbers, prior, alpha, accept);
// the two products with tables shared by all chains -- the proposal P = R'z (R the one pooled factor, dense d x d
// with its lower triangle zero: M[s*d + o] = R(s,o)) and the Gaussian target's y = Lam v -- run as MFMA tiles like in
// scam_pooled_kernel: B = the wave's own 64 chains' vector in LDS, A from the shared table, NB = 4 output blocks x
// 4 chain groups = 16 accumulators per pass.  Rows s beyond an output block's last column are zero in R and are
// skipped (exact: they would add 0*z).  The results come back to lane = chain order through LDS (P) or as the
// lane-local partial chains of ss (y).  Same arithmetic per chain as step_kernel<false,false,true>.
#ifndef MCX_POOLED_KU
#define MCX_POOLED_KU 4
#endif
#ifndef MCX_POOLED_CB
#define MCX_POOLED_CB 16
#endif
template <bool TRI>
MCX_DEV void mfma_wave_product(const double *__restrict__ M, const double *X, int lane, int d, int d4, int ob0, int nb,
                               mcx_d4 (&c)[4][4])
{
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) c[b][g] = mcx_d4{0.0, 0.0, 0.0, 0.0};
    int kmax = d4;
    if (TRI) { const int last = 16 * (ob0 + nb); kmax = last < d4 ? last : d4; }
    const double *__restrict__ ap = M + (size_t)lk * d + 16 * ob0 + li;
    const double *xp = X + lk * 64 + li;
    // KU k-blocks per trip, their 4 KU loads of the shared table first: a trip waits for the L2 once -- one k-block per trip put thirteen
    // round trips of ~1 us on each product of a wave that has the SIMD almost to itself (round 4: 0.93 -> 0.73 ms per iteration of 1 048
    // 576 chains at npar 50; two k-blocks per trip do almost as well, seven or eight are slower)
    constexpr int KU = MCX_POOLED_KU;
    for (int s0 = 0; s0 < kmax; s0 += 4 * KU) {
        double a[KU][4];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int s = (s0 + 4 * u < kmax) ? s0 + 4 * u : kmax - 4;          // (a k-block past the end: loaded again, not multiplied)
#pragma unroll
            for (int b = 0; b < 4; ++b) a[u][b] = ap[(size_t)s * d + 16 * (b < nb ? b : 0)];
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int s = s0 + 4 * u;
            if (s < kmax) {
                const double *xq = xp + s * 64;
                const double b0 = xq[0], b1 = xq[16], b2 = xq[32], b3 = xq[48];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (b < nb && (!TRI || s < 16 * (ob0 + b + 1))) {
                        c[b][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][b], b0, c[b][0], 0, 0, 0);
                        c[b][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][b], b1, c[b][1], 0, 0, 0);
                        c[b][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][b], b2, c[b][2], 0, 0, 0);
                        c[b][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][b], b3, c[b][3], 0, 0, 0);
                    }
                }
            }
        }
    }
}

// DR: delayed rejection's second stage on the same cores (drscale > 0): the stage-2 proposal with the shared R2 = R / drscale (g_R2T, dense
// like g_RT), the target once more, and the two quadratic forms dx' iC dx of MCMC_DR_alpha13 as y = iC dx products against the dense
// symmetric table g_iCd, each followed by the chain q = sum_i y_i dx_i ascending in i (lane = chain: y comes back through the LDS vector,
// dx waits in the chain's global scratch EngineDev::xscr).  Operation for operation step_body<false, true, true> (the lane-per-chain form
// with the tables through the scalar cache), whose chains these are. W2 (without delayed rejection): 256 registers, so that two waves share
// a SIMD (the LDS vector lets six waves on a CU at npar 50: two SIMDs with two).  The compiler spills ~40 doubles of state around the
// products to fit, and with more tiles than SIMDs it is still faster (round 4; round 2's attempt predates the single-pass LDS layout): 97.1
// -> 93.2 ms per 100 iterations of 1 048 576 chains at npar 50.  With one tile per SIMD or fewer there is nobody to share with and the
// spills are all it buys (npar 20, 65536 chains: 2.1e9 against 2.6e9 proposals/s): the host takes the 512-register instance there.
template <bool DR, bool W2 = false>
__global__ __launch_bounds__(64, (!DR && W2) ? 2 : 1) void pooled_mfma_kernel(EngineDev E, int it0, int it1,
                                                         const double *__restrict__ g_mu, const double *__restrict__ g_lamT,
                                                         const double *__restrict__ g_RT, const double *__restrict__ g_R2T,
                                                         const double *__restrict__ g_iCd)
{
    extern __shared__ double X[];
    const int lane = threadIdx.x, tile = blockIdx.x, d = E.d;
    const int d4 = (d + 3) & ~3, nt = (d + 15) >> 4, li = lane & 15, lk = lane >> 4;
    const bool single = (nt <= 4);                                  // one pass: the outputs may overwrite the input vector
    // single pass: the products (rows < d4 only) overwrite the vector they came from, and the partial ss chains go over
    // its first 4 nt rows once y = Lam v is in registers -- 512 d4 bytes of LDS per wave (26 KiB at d = 50: six waves per CU)
    double *T = single ? X : X + (size_t)d4 * 64;                  // [16 nt][64] products in (row, chain) order
    double *Q = single ? X : T + (size_t)nt * 16 * 64;             // [4 nt][64] partial ss chains
    double *theta_t = E.theta + (size_t)tile * d * 64;
    double *cand_t = E.cand + (size_t)tile * d * 64;
    double *c2_t = E.cs + (size_t)tile * 2 * d * 64;               // DR: the second-stage candidate
    double *xs_t = DR ? E.xscr + (size_t)tile * 2 * d * 64 : nullptr;   // DR: dx of the quadratic form in flight
    const bool gauss = (E.tgt.kind == TGT_GAUSS);
    LaneState L;
    lane_load(E, tile, lane, L);
    uint32_t erstayed = TIDX(E.ictr, tile, NICTR, I_ERSTAYED, lane);
    mcx_d4 c[4][4];
    // out = M' X on the matrix cores, into T in (row, chain) order; tri: M is upper triangular (rows beyond a block's last column are zero)
    auto product_to_T = [&](const double *__restrict__ M, bool tri) {
        for (int ob0 = 0; ob0 < nt; ob0 += 4) {
            const int nb = (nt - ob0) < 4 ? (nt - ob0) : 4;
            if (tri) mfma_wave_product<true>(M, X, lane, d, d4, ob0, nb, c);
            else mfma_wave_product<false>(M, X, lane, d, d4, ob0, nb, c);
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (b < nb) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * (ob0 + b) + lk + 4 * r;
                        if (row < d4) {                                 // rows >= d are never read
                            double *o = T + (size_t)row * 64 + li;
                            o[0] = c[b][0][r]; o[16] = c[b][1][r]; o[32] = c[b][2][r]; o[48] = c[b][3][r];
                        }
                    }
                }
        }
    };
    // ss of the Gaussian target for the vector v = x - mu in X (mcxt_ss_gauss): y = Lam v on the matrix cores, the partial chains
    // q over r of y v in the lanes that hold them, their sum per chain
    auto gauss_ss = [&]() -> double {
        for (int k = d; k < d4; ++k) XL(k) = 0.0;
        for (int ob0 = 0; ob0 < nt; ob0 += 4) {
            const int nb = (nt - ob0) < 4 ? (nt - ob0) : 4;
            mfma_wave_product<false>(g_lamT, X, lane, d, d4, ob0, nb, c);       // y = Lam v
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (b < nb) {
                    const int o0 = 16 * (ob0 + b) + lk;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {                               // q = chain over r of y v (mcxt_ss_gauss)
                        double q = c[b][g][0] * X[(size_t)o0 * 64 + 16 * g + li];
#pragma unroll
                        for (int r = 1; r < 4; ++r) {
                            const int o = o0 + 4 * r;
                            const double t = dfma(c[b][g][r], X[(size_t)(o < d4 ? o : 0) * 64 + 16 * g + li], q);
                            q = (o < d) ? t : q;
                        }
                        if (o0 < d) Q[(size_t)(4 * (ob0 + b) + lk) * 64 + 16 * g + li] = q;
                    }
                }
        }
        double ss = Q[lane];
#pragma unroll 4
        for (int e = 1; e < 4 * nt; ++e) if (16 * (e >> 2) + (e & 3) < d) ss = ss + Q[(size_t)e * 64 + lane];
        return ss;
    };
    // dst = theta + T (lane = chain), and v = dst - mu back into the LDS vector for the Gaussian target
    auto candidate_from_T = [&](double *dst_t) {
        constexpr int CB = MCX_POOLED_CB;                // state elements' loads before their stores (see copy_vec): sixteen in flight
        for (int k0 = 0; k0 < d; k0 += CB) {
            double th[CB], tv[CB];
#pragma unroll
            for (int u = 0; u < CB; ++u) { const int k = (k0 + u < d) ? k0 + u : d - 1; th[u] = GV(theta_t, k);
                tv[u] = T[(size_t)k * 64 + lane]; }
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                if (k0 + u < d) {
                    const double cnd = th[u] + tv[u];
                    GV(dst_t, k0 + u) = cnd;
                    if (gauss) XL(k0 + u) = cnd - g_mu[k0 + u];
                }
            }
        }
    };
    // -DMCX_PHASE_PROF (tools/build_variant.sh; profiles/r05_a/c4_pooled_phases.txt): where a wave's iteration goes, by wall_clock64
#ifdef MCX_PHASE_PROF
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = wall_clock64();
#define PH(i) { unsigned long long tn = wall_clock64(); ph[i] += tn - tq; tq = tn; }
#else
#define PH(i)
#endif
    for (int it = it0; it <= it1; ++it) {
        // ---- newpar = MCMC_propose(oldpar, R): z straight into the LDS vector, P = R'z on the matrix cores
        MCX_POOLED_GEN(L.g, X, lane, d, true);
        PH(0)
        if (it == it1) {                                               // the launch's last normals stay readable (pooled RAM statistic)
            double *zk = E.zs + ((size_t)tile * 2 + (it & 1)) * d * 64;
            for (int k = 0; k < d; ++k) GV(zk, k) = XL(k);
        }
        for (int k = d; k < d4; ++k) XL(k) = 0.0;
        product_to_T(g_RT, !E.usesvd);                                 // (condmax > 0: the full SVD factor)
        PH(1)
        candidate_from_T(cand_t);
        PH(2)
        bool inb = target_inbounds(E.tgt, d, lane, cand_t);
        double pri2 = target_prior(E.tgt, d, lane, cand_t);
        PH(3)
        double ss2 = gauss ? gauss_ss() : target_ss<false>(E.tgt, d, lane, cand_t, g_mu, g_lamT);
        PH(4)
        // ---- alpha, reject (MCMC_run.F90:47-63), as in step_kernel
        bool reject;
        if (!DR && E.method == M_ER) {                    // early rejection, MCMC_run_er.F90:60-89 (no second stage with it)
            if (!inb) { L.bnd += 1; reject = true; }
            else {
                double u = rng_uniform(L.g);              // MCMC_sscrit, MCMC_DRAM.F90:124-135: always drawn
                double sscrit = -2.0 * d_log(u) + L.ss1 / L.sigma2 + L.pri1;
                if (pri2 >= sscrit) { reject = true; erstayed += 1; }
                else { sscrit = L.sigma2 * (sscrit - pri2); reject = (ss2 >= sscrit); }
            }
        }
        else if (!inb) { if (!DR) L.bnd += 1; reject = true; L.alpha12 = 0.0; }
        else {
            L.alpha12 = d_alpha(L.ss1, L.pri1, ss2, pri2, L.sigma2);
            reject = true;
            if (L.alpha12 >= 1.0) reject = false;
            else if (L.alpha12 > 0.0) { double u = rng_uniform(L.g); if (u <= L.alpha12) reject = false; }
        }
        // ---- second stage: one delayed-rejection try with R2 = R/drscale (MCMC_run.F90:65-91)
        bool dr_moved = false;
        if (DR && __any(reject)) {
            const bool m = reject;
            if (m) L.drtries += 1;
            MCX_POOLED_GEN(L.g, X, lane, d, m);            // lanes that did not draw compute on stale values and are not looked at
            for (int k = d; k < d4; ++k) XL(k) = 0.0;
            product_to_T(g_R2T, !E.usesvd);
            candidate_from_T(c2_t);
            const bool inb2 = target_inbounds(E.tgt, d, lane, c2_t);
            const double pri3 = target_prior(E.tgt, d, lane, c2_t);
            const double ss3 = gauss ? gauss_ss() : target_ss<false>(E.tgt, d, lane, c2_t, g_mu, g_lamT);
            double qf[2];
#pragma unroll
            // qa: dx = newpar2 - newpar, qb: dx = oldpar - newpar (MCMC_DRAM.F90:180-182)
            for (int f = 0; f < 2; ++f) {
                const double *a_t = f == 0 ? c2_t : theta_t;
                for (int k = 0; k < d; ++k) { const double dx = GV(a_t, k) - GV(cand_t, k); XL(k) = dx; GV(xs_t, k) = dx; }
                for (int k = d; k < d4; ++k) XL(k) = 0.0;
                product_to_T(g_iCd, false);                            // y = iC dx
                double q = 0.0;
                for (int i = 0; i < d; ++i) q = q + T[(size_t)i * 64 + lane] * GV(xs_t, i);
                qf[f] = q;
            }
            if (m) {
                if (!inb2) L.bnd += 1;
                else {
                    // MCMC_DR_alpha13, MCMC_DRAM.F90:162-186
                    double alpha32;
                    if (L.alpha12 == 0.0) alpha32 = 0.0;
                    else alpha32 = min1(d_exp(-0.5 * ((ss2 - ss3) / L.sigma2 + (pri2 - pri3))));
                    const double l2 = -0.5 * ((ss3 - L.ss1) / L.sigma2 + (pri3 - L.pri1));
                    const double q1 = -0.5 * (qf[0] - qf[1]);
                    const double alpha13 = min1(d_exp(l2 + q1) * (1.0 - alpha32) / (1.0 - L.alpha12));
                    bool rej2 = true;
                    if (alpha13 >= 1.0) rej2 = false;
                    else if (alpha13 > 0.0) { double u = rng_uniform(L.g); if (u <= alpha13) rej2 = false; }
                    if (!rej2) { L.dracc += 1; reject = false; dr_moved = true; ss2 = ss3; pri2 = pri3; }
                }
            }
        }
        PH(5)
        if (reject) { L.stayed += 1; L.curcount += 1; }
        else { L.ss1 = ss2; L.pri1 = pri2; L.chainind += 1; L.curcount = 1; }
        if (E.updatesigma) {
            double gm = rng_gamma(L.g, E.gam_shape, 2.0 / (E.N0S02 + L.ss1));
            L.sigma2 = 1.0 / gm;
        }
        unsigned long long ballot = __ballot(!reject);
        const int slot = it % E.wcap;
        if (!reject) {
            double *h = E.hist ? E.hist + ((size_t)tile * E.wcap + slot) * (size_t)E.hs * 64 : nullptr;
            copy_vec_wide<MCX_POOLED_CB>(theta_t, dr_moved ? c2_t : cand_t, h, lane, d);   // newpar = newpar2 when the DR try was accepted
            if (h) GV(h, d) = L.ss1;
        }
        if (E.hist) {
            if (lane == 0) E.wacc[(size_t)tile * E.wcap + slot] = ballot;
            if (E.record_s2) E.s2hist[((size_t)tile * E.wcap + slot) * 64 + lane] = L.sigma2;
        }
        if (E.accmask && lane == 0) E.accmask[(size_t)(it - 1) * E.ntiles + tile] = ballot;
        PH(6)
    }
    lane_store(E, tile, lane, L);
    TIDX(E.ictr, tile, NICTR, I_ERSTAYED, lane) = erstayed;
#ifdef MCX_PHASE_PROF
    if (lane == 0 && (tile == 0 || tile == E.ntiles / 2 || tile == E.ntiles - 1))
        printf("pooled_mfma tile %d its %d x10ns: normals %llu product %llu candidate %llu bounds+prior %llu target %llu decide %llu accept+history %llu\n",
               tile, it1 - it0 + 1, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6]);
#endif
#undef PH
}

} // namespace mcx
