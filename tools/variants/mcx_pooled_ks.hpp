// tools/variants/mcx_pooled_ks.hpp -- a MEASURED NEGATIVE (round 6), not part of libmcmcx.so (tools/variants/README.md; -DMCX_VARIANTS, MCMCX_POOLED_KS=1).
// pooled_mfma_kernel<false, true> with the tile's LDS vector in two pieces of 40 rows, so that eight tiles fit a CU at npar 41..64 where the whole
// vector leaves six (npar 50: BASELINE config 4 pooled).  What two more tiles are worth was measured first, with the library's own kernel under an
// inflated LDS allocation: +26 % at npar 36 / 40 for eight tiles against six (tools/pooled_occupancy_probe.py, profiles/r06_i).  The kernel below is
// bit-equal to the library's on 77 configurations at its FIRST run (tools/pooled_ks_check.py: every npar 41..64 class, Gaussian / banana targets,
// bounds, priors, sigma2 update, early rejection, pooled RAM, SVD factor, burn-in, ragged tiles, cut runs) -- and SLOWER at config 4's size:
//   first form (six inlined copies of the product loop, 529 registers spilled)     136.6 ms per 100 iterations   (library: 63.3)
//   one product loop per product, runtime triangular flag (209 spilled)             81.9
//   + partial-ss chains of blocks 3, 2 first and v[0..12) restored to LDS, candidate batches of 8, two k-blocks per trip (169)   71.7   <- this file
//   + ONE generator call with per-access LDS / global destinations (instead of two calls)   81.1
// Eight tiles per CU buy +33 % waves; the two-piece form pays for them with five more dependent global round trips per iteration (the second
// piece of z, of v, v's restore, the candidate in two parts, the second generator chunk's parking), a second straggler wait in the generator
// and ~600 SGPR spill moves: 90 us per tile-iteration against 59.  profiles/r06_i/pooled_ks_check_*.txt.
#pragma once
namespace mcx {

// ---------------------------------------------------------------- the same with the tile's vector in TWO pieces: eight tiles per CU at npar 41..64 (round 6)
// pooled_mfma_kernel<false, true> holds the tile's whole vector in LDS -- (npar rounded up to four) rows of 512 bytes -- and that, not registers,
// caps the waves on a CU: eight up to npar 40, seven at 44, six at 50.  What two more tiles are worth was measured with the same kernel at the same
// npar under an inflated allocation (tools/pooled_occupancy_probe.py, profiles/r06_i): +26 % at npar 36 / 40 for eight tiles against six.
// Here the vector has PKS = 40 rows whatever npar is: both products run in two passes over k -- rows 0..39, then rows 40.. loaded into the same LDS
// rows -- into the SAME sixteen accumulators in the SAME ascending order of k-blocks, so every output is the chain of MFMAs pooled_mfma_kernel gives it:
//   normals     gen_normals_split twice: 40 deviates into LDS, the rest into the iteration's global scratch row (two calls consume the stream
//               exactly like one: a pair that straddles the cut leaves its second deviate cached, and the second call takes it first);
//   P = R'z     pass 1 over k < 40, rows 0.. <- z[40..], pass 2; outputs leave through the 40 rows in two pieces (blocks 2, 3 first, then 0, 1),
//               each followed by its part of candidate = theta + P (lane = chain) and, for the Gaussian target, v = candidate - mu in place;
//   y = Lam v   pass 1 over v[0..39], rows 0.. <- v[40..] = candidate - mu from the chain's global candidate, pass 2; the partial ss chains take
//               v from where it still stands (rows 40.. in LDS rows 0.., rows npar4-40..39 in place) or, for the rows the second piece
//               overwrote, from the global candidate again -- the same operands by the same operations; they are summed per chain in the order
//               of pooled_mfma_kernel once all of them exist (they wait in registers: their LDS rows are v's until then).
// No delayed rejection (the instance that shares a SIMD has none either).  Bit-equal to pooled_mfma_kernel<false, true> and the lane kernels
// (tests/test_gpu_pooled.py, tests/test_gpu_fuzz.py: every two-waves case at npar 41..64 runs both).
constexpr int PKS = 40;
MCX_DEV void mfma_wave_part(const double *__restrict__ M, const double *X, int xoff, int lane, int d, int nb, int s_lo, int s_hi,
                            mcx_d4 (&c)[4][4], bool TRI)
{
    const int li = lane & 15, lk = lane >> 4;
    int kmax = s_hi;
    if (TRI) { const int last = 16 * nb; kmax = last < kmax ? last : kmax; }
    const double *__restrict__ ap = M + (size_t)lk * d + li;
    const double *xp = X + lk * 64 + li;
    constexpr int KU = 2;
    for (int s0 = s_lo; s0 < kmax; s0 += 4 * KU) {
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
                const double *xq = xp + (s - xoff) * 64;
                const double b0 = xq[0], b1 = xq[16], b2 = xq[32], b3 = xq[48];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (b < nb && (!TRI || s < 16 * (b + 1))) {
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

__global__ __launch_bounds__(64, 2) void pooled_mfma_ks_kernel(EngineDev E, int it0, int it1, const double *__restrict__ g_mu,
                                                                const double *__restrict__ g_lamT, const double *__restrict__ g_RT)
{
    extern __shared__ double X[];                                   // [PKS][64]
    const int lane = threadIdx.x, tile = blockIdx.x, d = E.d;
    const int d4 = (d + 3) & ~3, nt = (d + 15) >> 4, li = lane & 15, lk = lane >> 4;      // 40 < d4 <= 64: nt = 3 or 4
    double *theta_t = E.theta + (size_t)tile * d * 64;
    double *cand_t = E.cand + (size_t)tile * d * 64;
    const bool gauss = (E.tgt.kind == TGT_GAUSS);
    const bool tri = !E.usesvd;
    LaneState L;
    lane_load(E, tile, lane, L);
    uint32_t erstayed = TIDX(E.ictr, tile, NICTR, I_ERSTAYED, lane);
    mcx_d4 c[4][4];
    auto zero_c = [&]() {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) c[b][g] = mcx_d4{0.0, 0.0, 0.0, 0.0};
    };
    // the accumulators of output blocks b0..b1-1 into LDS rows (their row - r0), in (row, chain) order
    auto blocks_to_rows = [&](int b0, int b1, int r0) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if (b >= b0 && b < b1 && b < nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * b + lk + 4 * r;
                    if (row < d4) {
                        double *o = X + (size_t)(row - r0) * 64 + li;
                        o[0] = c[b][0][r]; o[16] = c[b][1][r]; o[32] = c[b][2][r]; o[48] = c[b][3][r];
                    }
                }
            }
    };
    // candidate_k = theta_k + P_k for k0 <= k < k1 (P_k in LDS row k - r0), and v_k = candidate_k - mu_k into LDS row k where that row is v's
    auto candidate_part = [&](int k0, int k1, int r0) {
        constexpr int CB = 8;
        for (int kb = k0; kb < k1; kb += CB) {
            double th[CB], tv[CB];
#pragma unroll
            for (int u = 0; u < CB; ++u) { const int k = (kb + u < k1) ? kb + u : k1 - 1; th[u] = GV(theta_t, k);
                tv[u] = X[(size_t)(k - r0) * 64 + lane]; }
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                const int k = kb + u;
                if (k < k1) {
                    const double cnd = th[u] + tv[u];
                    GV(cand_t, k) = cnd;
                    if (gauss && k < PKS) XL(k) = cnd - g_mu[k];
                }
            }
        }
    };
    // c = M' x over both pieces of x: rows 0..PKS-1 stand in LDS; rows PKS.. come from the chain's global vector src (less mu for the target's v)
    auto two_piece_product = [&](const double *__restrict__ M, bool tr, const double *src, bool submu) {
        zero_c();
#pragma clang loop unroll(disable)
        for (int piece = 0; piece < 2; ++piece) {
            if (piece == 1) {
#pragma clang loop unroll(disable)
                for (int k = PKS; k < d4; ++k) {
                    double x = 0.0;
                    if (k < d) { x = GV(src, k); if (submu) x = x - g_mu[k]; }
                    XL(k - PKS) = x;
                }
            }
            mfma_wave_part(M, X, piece * PKS, lane, d, nt, piece * PKS, piece ? d4 : PKS, c, tr);
        }
    };
    for (int it = it0; it <= it1; ++it) {
        // ---- newpar = MCMC_propose(oldpar, R)
        double *zg = E.zs + ((size_t)tile * 2 + (it & 1)) * d * 64;    // this iteration's normals in global scratch: rows PKS.. always, all of them at the
        MCX_POOLED_GEN(L.g, X, lane, PKS, true);                       // launch's last iteration (pooled RAM statistic)
        MCX_POOLED_GEN(L.g, zg + (size_t)PKS * 64, lane, d - PKS, true);
        if (it == it1) for (int k = 0; k < PKS; ++k) GV(zg, k) = XL(k);
        two_piece_product(g_RT, tri, zg, false);
        blocks_to_rows(2, 4, 32);                                      // P[32..] through rows 0.. (rows 32..39 stay free for v)
        candidate_part(32, d, 32);
        blocks_to_rows(0, 2, 0);                                       // P[0..31] through rows 0..31, v[0..31] over them
        candidate_part(0, 32, 0);
        bool inb = target_inbounds(E.tgt, d, lane, cand_t);
        double pri2 = target_prior(E.tgt, d, lane, cand_t);
        double ss2;
        if (gauss) {                                                   // mcxt_ss_gauss: y = Lam v, the partial chains q over r of y v, their sum per chain
            two_piece_product(g_lamT, false, cand_t, true);
            const int nlost = d4 - PKS;                                // rows 0..nlost-1 of v were overwritten by the second piece
            double qv[4][4];
            // the chains of blocks 3, 2 first (their rows 40.. stand in LDS rows 0.., 32..39 in place); then v[0..nlost) comes back from the
            // chain's global candidate into its rows, and blocks 1, 0 read theirs
            auto block_q = [&](int b) {
                const int o0 = 16 * b + lk;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = 16 * g + li;
                    auto vrow = [&](int o) -> double {                 // v_o of chain ch
                        const int oc = o < d4 ? o : 0;
                        return X[(size_t)((b >= 2 && oc >= PKS) ? oc - PKS : oc) * 64 + ch];
                    };
                    double q = c[b][g][0] * vrow(o0);
#pragma unroll
                    for (int r = 1; r < 4; ++r) {
                        const int o = o0 + 4 * r;
                        const double t = dfma(c[b][g][r], vrow(o), q);
                        q = (o < d) ? t : q;
                    }
                    qv[b][g] = q;
                }
            };
            if (nt > 3) block_q(3);
            block_q(2);
#pragma clang loop unroll(disable)
            for (int k = 0; k < nlost; ++k) XL(k) = GV(cand_t, k) - g_mu[k];
            block_q(1);
            block_q(0);
            // every chain is read by now: the partial chains take rows 0..4 nt - 1 (the wave's LDS operations retire in order)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (b < nt && 16 * b + lk < d) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) X[(size_t)(4 * b + lk) * 64 + 16 * g + li] = qv[b][g];
                }
            double ss = X[lane];
#pragma unroll 4
            for (int e = 1; e < 4 * nt; ++e) if (16 * (e >> 2) + (e & 3) < d) ss = ss + X[(size_t)e * 64 + lane];
            ss2 = ss;
        } else ss2 = target_ss<false>(E.tgt, d, lane, cand_t, g_mu, g_lamT);
        // ---- alpha, reject (MCMC_run.F90:47-63), as in pooled_mfma_kernel
        bool reject;
        if (E.method == M_ER) {                           // early rejection, MCMC_run_er.F90:60-89
            if (!inb) { L.bnd += 1; reject = true; }
            else {
                double u = rng_uniform(L.g);              // MCMC_sscrit, MCMC_DRAM.F90:124-135: always drawn
                double sscrit = -2.0 * d_log(u) + L.ss1 / L.sigma2 + L.pri1;
                if (pri2 >= sscrit) { reject = true; erstayed += 1; }
                else { sscrit = L.sigma2 * (sscrit - pri2); reject = (ss2 >= sscrit); }
            }
        }
        else if (!inb) { L.bnd += 1; reject = true; L.alpha12 = 0.0; }
        else {
            L.alpha12 = d_alpha(L.ss1, L.pri1, ss2, pri2, L.sigma2);
            reject = true;
            if (L.alpha12 >= 1.0) reject = false;
            else if (L.alpha12 > 0.0) { double u = rng_uniform(L.g); if (u <= L.alpha12) reject = false; }
        }
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
            copy_vec_wide<MCX_POOLED_CB>(theta_t, cand_t, h, lane, d);
            if (h) GV(h, d) = L.ss1;
        }
        if (E.hist) {
            if (lane == 0) E.wacc[(size_t)tile * E.wcap + slot] = ballot;
            if (E.record_s2) E.s2hist[((size_t)tile * E.wcap + slot) * 64 + lane] = L.sigma2;
        }
        if (E.accmask && lane == 0) E.accmask[(size_t)(it - 1) * E.ntiles + tile] = ballot;
    }
    lane_store(E, tile, lane, L);
    TIDX(E.ictr, tile, NICTR, I_ERSTAYED, lane) = erstayed;
}


} // namespace mcx
