// mcx_pooled_ks.hpp -- pooled_mfma_ks_kernel: pooled_mfma_kernel<false, true> with the tile's LDS vector in TWO pieces of forty rows, so
// that eight tiles fit a CU at npar 41..64 where the whole vector leaves seven to five (npar 50, BASELINE config 4 pooled: six). Round 6.
// pooled_mfma_kernel holds the tile's whole vector in LDS -- (npar rounded up to four) rows of 512 bytes -- and that, not registers, caps
// the waves on a CU. What two more tiles are worth was measured first, with that kernel at npar 36 / 40 under an inflated allocation
// (tools/pooled_occupancy_probe.py, profiles/r06_i): +26 % for eight tiles against six.
// Here the vector has PKS = 40 rows whatever npar is: both products run in two passes over k -- rows 0..39, then rows 40.. loaded into
// the same LDS rows -- into the SAME sixteen accumulators in the SAME ascending order of k-blocks, so every output is the chain of MFMAs
// pooled_mfma_kernel gives it:
//   normals gen_normals_split twice: 40 deviates into LDS, the rest into the iteration's half of the chain's global pair of normal
//              vectors (two calls consume the stream exactly like one: a pair that straddles the cut leaves its second deviate cached,
//              and the second call takes it first);
//   P = R'z pass 1 over k < 40, LDS rows 0.. <- z[40..], pass 2; the sixteen accumulators leave at once: rows < 40 into the LDS rows,
//              rows 40.. into the OTHER half of the global pair; candidate = theta + P (lane = chain) in two loops, one per address
//              space, and for the Gaussian target v = candidate - mu in place for the rows LDS holds;
//   y = Lam v pass 1 over v[0..39], LDS rows 0.. <- v[40..] = candidate - mu from the chain's global candidate, pass 2; the partial ss
//              chains of blocks 3, 2 read v where it stands (rows 40.. in LDS rows 0.., rows 32..39 in place), then v[0..npar4-40) --
//              asked for before them -- goes back into its rows for blocks 1, 0: the same operands by the same operations; the chains
//              wait in registers (their LDS rows are v's until then) and are summed per chain in the order of pooled_mfma_kernel.
// No delayed rejection (pooled_mfma_kernel<true> keeps it). Bit-equal to pooled_mfma_kernel<false, true> and the lane kernels:
// tests/test_gpu_pooled.py (every class of npar 41..64 both ways), tools/pooled_ks_check.py (77 configurations).
// How it got from 136.6 ms per 100 iterations at config 4's size to 59.2 (pooled_mfma_kernel<false, true>: 63.5) -- docs/history/r06.md
// section 8c:
//   136.6  six inlined copies of the product loop (their hoisted address arithmetic live across the iteration: 529 registers spilled)
//    81.9  one loop per product, a runtime triangular flag
//    71.7  the partial-ss chains of blocks 3, 2 first and v[0..12) restored to LDS
//    67.3  the second piece's reloads in batches of twelve (an `unroll(disable)` loop had serialised twelve dependent round trips)
//    65.3  all sixteen accumulators out at once (LDS + the other global half), ONE candidate pass
//    59.2  that pass as two loops, one per address space (`k < 40 ? LDS : global` per access had compiled to sixteen flat loads)
//    54.5  npar 49..52: the fourth output block's four rows through v_mfma_f64_4x4x4_4b_f64 (mfma_wave_part<S3>)
// Measured and not kept: one generator call with per-access destinations (81.1), the second piece prefetched into registers before pass 1
// (105 .. 118: no room beside sixteen accumulators), the second piece and v's restore by global_load_lds from rows parked in global
// (60.5: the extra stores cost more than the two hidden round trips).
#pragma once
namespace mcx {

constexpr int PKS = 40;
#ifndef MCX_KS_KU
#define MCX_KS_KU 4           // k-blocks whose A operands are asked for together
#endif
#ifndef MCX_KS_CL
#define MCX_KS_CL 20          // the candidate pass' batch over the LDS rows (divides PKS) and over the global rows
#define MCX_KS_CG 12
#endif
// S3 (npar 49..52: the fourth output block has FOUR rows): that block through v_mfma_f64_4x4x4_4b_f64 -- a quarter of the passes of a
// 16 x 16 x 4.  Its operands sit where the large instruction's do (k = lane / 16 in A and B; tools/mfma4_probe.hip, profiles/r06_m): B
// is the SAME register (chains 16 g + lane % 16), A the element of row 48 + lane % 4 instead of 48 + lane % 16, and D lane (lane / 16,
// lane % 16) holds row 48 + lane / 16 of chain 16 g + lane % 16 -- component 0 of the large instruction's accumulator, which is all of
// it that rows < 52 ever used.  The four products are added to C in ascending k, one fma each, like the large instruction's: the same
// bits (77 configurations; 57.7 -> 54.5 ms per 100 iterations at config 4's size).
template <bool S3>
MCX_DEV void mfma_wave_part(const double *__restrict__ M, const double *X, int xoff, int lane, int d, int nb, int s_lo, int s_hi,
                            mcx_d4 (&c)[4][4], double (&c3)[4], bool TRI)
{
    const int li = lane & 15, lk = lane >> 4;
    const double *__restrict__ ap4 = M + (size_t)lk * d + 48 + (li & 3);
    int kmax = s_hi;
    if (TRI) { const int last = 16 * nb; kmax = last < kmax ? last : kmax; }
    const double *__restrict__ ap = M + (size_t)lk * d + li;
    const double *xp = X + lk * 64 + li;
    constexpr int KU = MCX_KS_KU;
    for (int s0 = s_lo; s0 < kmax; s0 += 4 * KU) {
        double a[KU][4];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int s = (s0 + 4 * u < kmax) ? s0 + 4 * u : kmax - 4;          // (a k-block past the end: loaded again, not multiplied)
#pragma unroll
            for (int b = 0; b < 4; ++b) a[u][b] = (S3 && b == 3) ? ap4[(size_t)s * d] : ap[(size_t)s * d + 16 * (b < nb ? b : 0)];
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int s = s0 + 4 * u;
            if (s < kmax) {
                const double *xq = xp + (s - xoff) * 64;
                const double b0 = xq[0], b1 = xq[16], b2 = xq[32], b3 = xq[48];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (S3 && b == 3) {
                        c3[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[u][3], b0, c3[0], 0, 0, 0);
                        c3[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[u][3], b1, c3[1], 0, 0, 0);
                        c3[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[u][3], b2, c3[2], 0, 0, 0);
                        c3[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[u][3], b3, c3[3], 0, 0, 0);
                    } else if (b < nb && (!TRI || s < 16 * (b + 1))) {
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

template <bool S3>
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
    double c3[4];                                                   // S3: the fourth block's accumulators (c[3] is never touched then)
    auto zero_c = [&]() {
#pragma unroll
        for (int b = 0; b < (S3 ? 3 : 4); ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) c[b][g] = mcx_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int g = 0; g < 4; ++g) c3[g] = 0.0;
    };
    // the sixteen accumulators leave at once, in (row, chain) order: rows < PKS into the LDS rows, rows PKS.. into the iteration's global
    // scratch
    // row (the second piece of z that stood there is dead) -- which of the two is a compile-time matter (16 b + 4 r against PKS: lk < 4)
    auto outputs_to_rows = [&](double *zg) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if (b < nt) {
#pragma unroll
                for (int r = 0; r < ((S3 && b == 3) ? 1 : 4); ++r) {
                    const int row = 16 * b + lk + 4 * r;
                    // (rows d..d4-1 do not exist in the global vector; LDS rows are all < PKS < d)
                    if (row < d) {
                        double *o = ((16 * b + 4 * r < PKS) ? X : zg) + (size_t)row * 64 + li;
                        if (S3 && b == 3) { o[0] = c3[0]; o[16] = c3[1]; o[32] = c3[2]; o[48] = c3[3]; }
                        else { o[0] = c[b][0][r]; o[16] = c[b][1][r]; o[32] = c[b][2][r]; o[48] = c[b][3][r]; }
                    }
                }
            }
    };
    // candidate_k = theta_k + P_k (lane = chain; P_k from LDS row k below PKS, from the global row beyond), v_k = candidate_k - mu_k over
    // P's LDS rows for the Gaussian target
    auto candidate_all = [&](const double *zg) {
        // two loops, so that each access knows its address space (one loop with a per-k choice compiles to flat loads)
        constexpr int CL = MCX_KS_CL, CG = MCX_KS_CG;
#pragma clang loop unroll(disable)
        for (int kb = 0; kb < PKS; kb += CL) {
            double th[CL], tv[CL];
#pragma unroll
            for (int u = 0; u < CL; ++u) { th[u] = GV(theta_t, kb + u); tv[u] = X[(size_t)(kb + u) * 64 + lane]; }
#pragma unroll
            for (int u = 0; u < CL; ++u) {
                const double cnd = th[u] + tv[u];
                GV(cand_t, kb + u) = cnd;
                if (gauss) XL(kb + u) = cnd - g_mu[kb + u];
            }
        }
#pragma clang loop unroll(disable)
        for (int kb = PKS; kb < d; kb += CG) {
            double th[CG], tv[CG];
#pragma unroll
            for (int u = 0; u < CG; ++u) { const int k = (kb + u < d) ? kb + u : d - 1; th[u] = GV(theta_t, k);
                tv[u] = zg[(size_t)k * 64 + lane]; }
#pragma unroll
            for (int u = 0; u < CG; ++u)
                if (kb + u < d) GV(cand_t, kb + u) = th[u] + tv[u];
        }
    };
    // c = M' x over both pieces of x: rows 0..PKS-1 stand in LDS; rows PKS.. come from the chain's global vector src (less mu for the
    // target's v)
    auto two_piece_product = [&](const double *__restrict__ M, bool tr, const double *src, bool submu) {
        zero_c();
        // ONE copy of the k loop, run twice (six inlined copies had their address arithmetic hoisted and live across the iteration)
#pragma clang loop unroll(disable)
        for (int piece = 0; piece < 2; ++piece) {
            if (piece == 1) {
                // (twelve loads in flight per batch: one round trip for config 4's twelve rows, not twelve)
#pragma clang loop unroll(disable)
                for (int k0 = PKS; k0 < d4; k0 += 12) {
                    double xv[12];
#pragma unroll
                    for (int u = 0; u < 12; ++u) { const int k = k0 + u; xv[u] = GV(src, k < d ? k : d - 1); }
#pragma unroll
                    for (int u = 0; u < 12; ++u) {
                        const int k = k0 + u;
                        if (k < d4) XL(k - PKS) = (k < d) ? (submu ? xv[u] - g_mu[k] : xv[u]) : 0.0;
                    }
                }
            }
            mfma_wave_part<S3>(M, X, piece * PKS, lane, d, nt, piece * PKS, piece ? d4 : PKS, c, c3, tr);
        }
    };
#ifdef MCX_PHASE_PROF
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = wall_clock64();
#define PHK(i) { unsigned long long tn = wall_clock64(); ph[i] += tn - tq; tq = tn; }
#else
#define PHK(i)
#endif
    for (int it = it0; it <= it1; ++it) {
        // ---- newpar = MCMC_propose(oldpar, R)
        // this iteration's normals in global scratch: rows PKS.. always, all of them at the launch's last iteration (pooled RAM statistic)
        double *zg = E.zs + ((size_t)tile * 2 + (it & 1)) * d * 64;
        MCX_POOLED_GEN(L.g, X, lane, PKS, true);
        MCX_POOLED_GEN(L.g, zg + (size_t)PKS * 64, lane, d - PKS, true);
        if (it == it1) for (int k = 0; k < PKS; ++k) GV(zg, k) = XL(k);
        PHK(0)
        two_piece_product(g_RT, tri, zg, false);
        PHK(1)
        {   // (the OTHER half of the chain's two normal vectors takes P's rows PKS..: this half keeps z for the pooled RAM statistic)
            double *po = E.zs + ((size_t)tile * 2 + ((it + 1) & 1)) * d * 64;
            outputs_to_rows(po);
            candidate_all(po);
        }
        PHK(2)
        bool inb = target_inbounds(E.tgt, d, lane, cand_t);
        double pri2 = target_prior(E.tgt, d, lane, cand_t);
        PHK(3)
        double ss2;
        // mcxt_ss_gauss: y = Lam v, the partial chains q over r of y v, their sum per chain
        if (gauss) {
            two_piece_product(g_lamT, false, cand_t, true);
            PHK(7)
            const int nlost = d4 - PKS;                                // rows 0..nlost-1 of v were overwritten by the second piece
            double qv[4][4];
            // the chains of blocks 3, 2 first (their rows 40.. stand in LDS rows 0.., 32..39 in place); then v[0..nlost) comes back from
            // the chain's global candidate into its rows, and blocks 1, 0 read theirs
            auto block_q = [&](int b) {
                const int o0 = 16 * b + lk;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = 16 * g + li;
                    auto vrow = [&](int o) -> double {                 // v_o of chain ch
                        const int oc = o < d4 ? o : 0;
                        return X[(size_t)((b >= 2 && oc >= PKS) ? oc - PKS : oc) * 64 + ch];
                    };
                    double q = ((S3 && b == 3) ? c3[g] : c[b][g][0]) * vrow(o0);
#pragma unroll
                    for (int r = 1; r < ((S3 && b == 3) ? 1 : 4); ++r) {
                        const int o = o0 + 4 * r;
                        const double t = dfma(c[b][g][r], vrow(o), q);
                        q = (o < d) ? t : q;
                    }
                    qv[b][g] = q;
                }
            };
            // v[0..12) is asked for before the chains of blocks 3, 2 and written behind them: its round trip hides behind their LDS reads
            double xr[12];
#pragma unroll
            for (int u = 0; u < 12; ++u) xr[u] = GV(cand_t, u);
            if (nt > 3) block_q(3);
            block_q(2);
#pragma unroll
            for (int u = 0; u < 12; ++u) if (u < nlost) XL(u) = xr[u] - g_mu[u];
#pragma clang loop unroll(disable)
            for (int k0 = 12; k0 < nlost; k0 += 12) {
                double xv[12];
#pragma unroll
                for (int u = 0; u < 12; ++u) xv[u] = GV(cand_t, k0 + u);       // (k0 + u < 24 <= npar)
#pragma unroll
                for (int u = 0; u < 12; ++u) if (k0 + u < nlost) XL(k0 + u) = xv[u] - g_mu[k0 + u];
            }
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
        PHK(4)
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
        PHK(5)
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
        PHK(6)
    }
    lane_store(E, tile, lane, L);
    TIDX(E.ictr, tile, NICTR, I_ERSTAYED, lane) = erstayed;
#ifdef MCX_PHASE_PROF
    if (lane == 0 && (tile == 0 || tile == E.ntiles / 2 || tile == E.ntiles - 1))
        printf("pooled_mfma_ks tile %d its %d x10ns: normals %llu product %llu candidate %llu bounds+prior %llu target %llu (of it y = Lam v: %llu) decide %llu accept+history %llu\n",
               tile, it1 - it0 + 1, ph[0], ph[1], ph[2], ph[3], ph[4] + ph[7], ph[7], ph[5], ph[6]);
#endif
#undef PHK
}


} // namespace mcx
