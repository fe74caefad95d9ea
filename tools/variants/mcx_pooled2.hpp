// tools/variants/mcx_pooled2.hpp -- a MEASURED NEGATIVE, not part of libmcmcx.so (tools/variants/README.md; included by variants.inc under -DMCX_VARIANTS).
// Pooled AM / RAM / early rejection on the matrix cores with TWO WAVES PER TILE (round 5).
//
// pooled_mfma_kernel (mcx_kernels.hpp) runs one wave per tile of 64 chains.  Its iteration is half random numbers (VALU issue) and half
// latency -- operand trips of the two products, the state's round trips, LDS transposes (profiles/r05_a/c4_pooled_phases.txt) -- and the one
// thing that would hide the latency, more waves on the SIMD, is capped by the tile's LDS vector (52 rows x 512 B = 26.6 kB at npar 50: six
// tiles on a CU, 1.5 waves per SIMD).  Here a tile is a WORKGROUP OF TWO WAVES sharing that one vector: six tiles on a CU are twelve waves, three per
// SIMD, and every phase of the iteration is split between the two:
//   normals      wave w draws for the chains 32 w .. 32 w + 31, TWO LANES PER CHAIN (lane l and l + 32): Philox is counter-based, so the lanes of a
//                pair compute the attempts 0..NBL-1 and NBL..2 NBL-1 of the chain's next 2 NBL polar attempts side by side; one exchanged bit mask
//                tells both which attempts were accepted, a prefix count puts every accepted pair where normal_bm's one-at-a-time loop would have
//                put it, attempts behind the one that completes the vector are dropped with their uniforms undrawn (mcmcrand.F90:166-190: stream
//                position, deviates and the cached second deviate are the reference's -- gen_normals / group_normals use the same argument);
//   products     wave w owns two of the (at most four) 16-row output blocks of P = R'z and of y = Lam v for all four chain groups -- for the
//                triangular factor the blocks {0, 3} and {1, 2}, which balances their k-ranges -- sixteen-by-sixteen f64 MFMA tiles accumulated
//                ascending in k exactly as pooled_mfma_kernel does (the chain of fmas of one output never changes hands);
//   lane = chain candidate, accept copy and history row: the two lanes of a chain take one half of the parameters each; everything scalar per chain
//                (prior, MCMC_alpha, MCMC_reject's uniform, the sigma2 update, counters) is computed by BOTH lanes on identical inputs, so the
//                chain's state -- stream position included -- stays identical in the two without an exchange.
// Six workgroup barriers per iteration (two waves: cheap).  Same arithmetic per chain as pooled_mfma_kernel<false> and step_kernel<false, false, true>:
// bit-equal (tests/test_gpu_pooled.py, tests/test_gpu_fuzz.py: every pooled case on all three).  Covers 17 <= npar <= 64 (two to four output blocks,
// single LDS pass), no delayed rejection; the one-wave kernels keep the rest.
#pragma once

namespace mcx {

#ifndef MCX_POOLED2_NBL
#define MCX_POOLED2_NBL 4          // polar attempts per lane and trip (eight per chain and trip, like pooled_mfma_kernel's)
#endif

// z <- the chain's next d deviates into the chain's column of the LDS vector; h: which of the chain's two lanes this is.
// `participate` and the chain's stream state g are identical in the two lanes on entry, and on exit.
// (Xc: the chain's column of the LDS vector, xs: its row stride in doubles)
template <int NBL>
MCX_DEV void gen_normals2(Rng &g, double *Xc, int xs, int h, int d, bool participate)
{
    int k = 0;
    if (participate && g.saved && d > 0) { if (h == 0) Xc[0] = g.saved_y; g.saved = 0; k = 1; }     // normal_bm's cached deviate, mcmcrand.F90:172-175
    bool need = participate && (k < d);
    while (__any(need)) {
        const uint64_t b0 = (g.n >> 1) + (uint64_t)(h * NBL);
        const bool odd = (g.n & 1) != 0;
        uint32_t w[NBL + 1][4];
#pragma unroll
        for (int j = 0; j < NBL; ++j) philox4x32_10((uint32_t)(b0 + j), (uint32_t)((b0 + j) >> 32), g.k0, g.k1, w[j][0], w[j][1], w[j][2], w[j][3]);
        if (__any(need && odd)) philox4x32_10((uint32_t)(b0 + NBL), (uint32_t)((b0 + NBL) >> 32), g.k0, g.k1, w[NBL][0], w[NBL][1], w[NBL][2], w[NBL][3]);
        else { w[NBL][0] = w[NBL][1] = w[NBL][2] = w[NBL][3] = 0u; }
        double za[NBL], zb[NBL];
        unsigned okm = 0u;
#pragma unroll
        for (int j = 0; j < NBL; ++j) {
            // uniforms 2 (n/2 + a) and the next one of attempt a = h NBL + j (random_number(x), x(2): mcmcrand.F90:177)
            double x1 = odd ? bits_to_uniform(w[j][2], w[j][3]) : bits_to_uniform(w[j][0], w[j][1]);
            double x2 = odd ? bits_to_uniform(w[j + 1][0], w[j + 1][1]) : bits_to_uniform(w[j][2], w[j][3]);
            x1 = 2.0 * x1 - 1.0; x2 = 2.0 * x2 - 1.0;
            const double xx = x1 * x1 + x2 * x2;
            const bool ok = (xx < 1.0) && (xx != 0.0);
            const double z = sqrt(-2.0 * d_log(ok ? xx : 0.5) / (ok ? xx : 0.5));
            zb[j] = z * x1; za[j] = z * x2;
            okm |= ok ? (1u << j) : 0u;
        }
        // the pair's attempts in stream order: the first lane's, then the second's
        const unsigned other = (unsigned)__shfl_xor((int)okm, 32);
        const unsigned all = h ? (other | (okm << NBL)) : (okm | (other << NBL));
        double mysave = 0.0;
        int done_at = -1;                                   // attempt that completes the vector (this trip), -1: none
        if (need) {
            const int m = (d - k + 1) >> 1;                 // accepted pairs the chain still needs
            const int tot = __popc(all);
#pragma unroll
            for (int j = 0; j < NBL; ++j) {
                const int a = h * NBL + j;
                const int pre = __popc(all & ((1u << a) - 1u));              // accepted attempts before this one
                if (((okm >> j) & 1u) && pre < m) {
                    const int pos = k + 2 * pre;
                    Xc[(size_t)pos * xs] = za[j];
                    if (pos + 1 < d) Xc[(size_t)(pos + 1) * xs] = zb[j]; else mysave = zb[j];
                }
            }
            if (tot >= m) {
                int cnt = 0;
#pragma unroll
                for (int a = 0; a < 2 * NBL; ++a) { cnt += (int)((all >> a) & 1u); if (cnt == m && done_at < 0) done_at = a; }
                g.n += 2ull * (uint64_t)(done_at + 1);      // attempts up to and including the one that completes the vector
            } else {
                g.n += 2ull * (uint64_t)(2 * NBL);
                k += 2 * tot;
            }
            g.cblk = 0;                                     // the half-used block (n odd) is recomputed by the next single draw
        }
        // the second deviate of the completing attempt is the chain's new cache when an odd number was missing: its lane tells the other
        const double othersave = __shfl_xor(mysave, 32);
        if (need && done_at >= 0) {
            if (((d - k) & 1) != 0) { g.saved = 1; g.saved_y = ((done_at / NBL) == h) ? mysave : othersave; }
            k = d; need = false;
        }
    }
}

// out(16 blk + i, chain) = sum_s M[s * d + 16 blk + i] X[s][chain] for the (at most two) output blocks blk0, blk1 (-1: none) and all four chain
// groups, k-blocks of four ascending; TRI: M is upper triangular (rows s >= 16 (blk + 1) are zero and skipped, as in mfma_wave_product)
template <bool TRI>
MCX_DEV void mfma_product_blocks(const double *__restrict__ M, const double *X, int pl, int d, int d4, int blk0, int blk1, mcx_d4 (&c)[2][4])
{
    const int li = pl & 15, lk = pl >> 4;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) c[b][g] = mcx_d4{0.0, 0.0, 0.0, 0.0};
    const int e0 = TRI ? (16 * (blk0 + 1) < d4 ? 16 * (blk0 + 1) : d4) : d4;
    const int e1 = blk1 < 0 ? 0 : (TRI ? (16 * (blk1 + 1) < d4 ? 16 * (blk1 + 1) : d4) : d4);
    const int kmax = e0 > e1 ? e0 : e1;
    const double *__restrict__ ap0 = M + (size_t)lk * d + 16 * blk0 + li;
    const double *__restrict__ ap1 = M + (size_t)lk * d + 16 * (blk1 < 0 ? blk0 : blk1) + li;
    const double *xp = X + lk * 64 + li;
    constexpr int KU = MCX_POOLED_KU;
    for (int s0 = 0; s0 < kmax; s0 += 4 * KU) {
        double a[KU][2];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int s = (s0 + 4 * u < kmax) ? s0 + 4 * u : kmax - 4;          // (a k-block past the end: loaded again, not multiplied)
            a[u][0] = ap0[(size_t)s * d]; a[u][1] = ap1[(size_t)s * d];
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int s = s0 + 4 * u;
            if (s < kmax) {
                const double *xq = xp + s * 64;
                const double b0 = xq[0], b1 = xq[16], b2 = xq[32], b3 = xq[48];
                if (s < e0) {
                    c[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][0], b0, c[0][0], 0, 0, 0);
                    c[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][0], b1, c[0][1], 0, 0, 0);
                    c[0][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][0], b2, c[0][2], 0, 0, 0);
                    c[0][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][0], b3, c[0][3], 0, 0, 0);
                }
                if (s < e1) {
                    c[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][1], b0, c[1][0], 0, 0, 0);
                    c[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][1], b1, c[1][1], 0, 0, 0);
                    c[1][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][1], b2, c[1][2], 0, 0, 0);
                    c[1][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][1], b3, c[1][3], 0, 0, 0);
                }
            }
        }
    }
}

#ifndef MCX_POOLED2_WAVES
#define MCX_POOLED2_WAVES 3        // waves per SIMD asked of the register allocator (168 registers)
#endif
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(MCX_POOLED2_WAVES, MCX_POOLED2_WAVES)))
void pooled_mfma2_kernel(EngineDev E, int it0, int it1, const double *__restrict__ g_mu, const double *__restrict__ g_lamT, const double *__restrict__ g_RT)
{
    extern __shared__ double X[];                        // the tile's vector [d4][64] (single pass: products and partial ss chains reuse its rows)
    __shared__ uint32_t sbal[2];                         // the two waves' halves of the tile's accept ballot
    const int tid = threadIdx.x, w = tid >> 6, pl = tid & 63, tile = blockIdx.x, d = E.d;
    const int lane = 32 * w + (pl & 31), h = pl >> 5;    // lane: the chain's index in the tile (what GV / XL / TIDX index by); h: which of its two lanes
    const int d4 = (d + 3) & ~3, nt = (d + 15) >> 4, li = pl & 15, lk = pl >> 4;
    // this wave's output blocks: {0, nt - 1} and the middle ones (triangular k-ranges 16, 32, 48, d4: 16 + d4 against 32 + 48); two blocks: one each
    const int blk0 = (nt == 2) ? w : (w == 0 ? 0 : 1);
    const int blk1 = (nt == 2) ? -1 : (w == 0 ? nt - 1 : (nt == 4 ? 2 : -1));
    const int kh0 = h ? (d + 1) / 2 : 0, kh1 = h ? d : (d + 1) / 2;            // this lane's half of the chain's parameters
    double *theta_t = E.theta + (size_t)tile * d * 64;
    double *cand_t = E.cand + (size_t)tile * d * 64;
    const bool gauss = (E.tgt.kind == TGT_GAUSS);
    const bool plain = gauss && !E.tgt.lo && !E.tgt.hi && !E.tgt.pmu;            // nothing reads the candidate's other half through global memory
    LaneState L;
    lane_load(E, tile, lane, L);
    uint32_t erstayed = TIDX(E.ictr, tile, NICTR, I_ERSTAYED, lane);
    mcx_d4 c[2][4];
#ifdef MCX_PHASE_PROF
    unsigned long long ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tq = wall_clock64();
#define PH(i) { unsigned long long tn = wall_clock64(); ph[i] += tn - tq; tq = tn; }
#else
#define PH(i)
#endif
    constexpr int CBH = 16;                              // state elements per batch (a lane moves half of the chain's: two batches at npar 50)
    for (int it = it0; it <= it1; ++it) {
        // ---- newpar = MCMC_propose(oldpar, R): z straight into the LDS vector, P = R'z on the matrix cores
        gen_normals2<MCX_POOLED2_NBL>(L.g, X + lane, 64, h, d, true);
        if (h == 0) for (int k = d; k < d4; ++k) XL(k) = 0.0;
        PH(0)
        __syncthreads();                                                          // (a) the 64 chains' normals are in X
        PH(1)
        if (it == it1) {                                                          // the launch's last normals stay readable (pooled RAM statistic)
            double *zk = E.zs + ((size_t)tile * 2 + (it & 1)) * d * 64;
            for (int k = kh0; k < kh1; ++k) GV(zk, k) = XL(k);
        }
        if (E.usesvd) mfma_product_blocks<false>(g_RT, X, pl, d, d4, blk0, blk1, c);   // (condmax > 0: the full SVD factor)
        else mfma_product_blocks<true>(g_RT, X, pl, d, d4, blk0, blk1, c);
        PH(2)
        __syncthreads();                                                          // (b) both waves have read the normals: the products may overwrite them
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int blk = b == 0 ? blk0 : blk1;
            if (blk >= 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * blk + lk + 4 * r;
                    if (row < d4) { double *o = X + (size_t)row * 64 + li; o[0] = c[b][0][r]; o[16] = c[b][1][r]; o[32] = c[b][2][r]; o[48] = c[b][3][r]; }
                }
            }
        }
        __syncthreads();                                                          // (c) P in (row, chain) order
        PH(3)
        for (int k0 = kh0; k0 < kh1; k0 += CBH) {         // cand = theta + P for this lane's half; v = cand - mu back into the LDS vector for the Gaussian target
            double th[CBH], tv[CBH];
#pragma unroll
            for (int u = 0; u < CBH; ++u) { const int k = (k0 + u < kh1) ? k0 + u : kh1 - 1; th[u] = GV(theta_t, k); tv[u] = XL(k); }
#pragma unroll
            for (int u = 0; u < CBH; ++u) {
                if (k0 + u < kh1) {
                    const double cnd = th[u] + tv[u];
                    GV(cand_t, k0 + u) = cnd;
                    if (gauss) XL(k0 + u) = cnd - g_mu[k0 + u];
                }
            }
        }
        PH(4)
        if (!plain) { __threadfence_block(); __syncthreads(); }                   // bounds / prior / a non-Gaussian target read the whole candidate from global memory
        const bool inb = target_inbounds(E.tgt, d, lane, cand_t);
        const double pri2 = target_prior(E.tgt, d, lane, cand_t);
        double ss2;
        if (gauss) {
            if (h == 0) for (int k = d; k < d4; ++k) XL(k) = 0.0;
            __syncthreads();                                                      // (d) v = cand - mu of all 64 chains
            PH(5)
            mfma_product_blocks<false>(g_lamT, X, pl, d, d4, blk0, blk1, c);     // y = Lam v
            double q[2][4];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int blk = b == 0 ? blk0 : blk1;
                const int o0 = 16 * (blk < 0 ? 0 : blk) + lk;
#pragma unroll
                for (int g = 0; g < 4; ++g) {                                     // q = chain over r of y v (mcxt_ss_gauss)
                    double qq = c[b][g][0] * X[(size_t)o0 * 64 + 16 * g + li];
#pragma unroll
                    for (int r = 1; r < 4; ++r) {
                        const int o = o0 + 4 * r;
                        const double t = dfma(c[b][g][r], X[(size_t)(o < d4 ? o : 0) * 64 + 16 * g + li], qq);
                        qq = (o < d) ? t : qq;
                    }
                    q[b][g] = qq;
                }
            }
            PH(6)
            __syncthreads();                                                      // (e) both waves have read v: the partial chains may overwrite its first rows
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int blk = b == 0 ? blk0 : blk1;
                if (blk >= 0 && 16 * blk + lk < d) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) X[(size_t)(4 * blk + lk) * 64 + 16 * g + li] = q[b][g];
                }
            }
            __syncthreads();                                                      // (f)
            PH(7)
            ss2 = XL(0);
#pragma unroll 4
            for (int e = 1; e < 4 * nt; ++e) if (16 * (e >> 2) + (e & 3) < d) ss2 = ss2 + XL(e);
        } else {
            ss2 = target_ss<false>(E.tgt, d, lane, cand_t, g_mu, g_lamT);
        }
        // ---- alpha, reject (MCMC_run.F90:47-63; MCMC_run_er.F90:60-89), as in pooled_mfma_kernel -- by both lanes of the chain
        bool reject;
        if (E.method == M_ER) {
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
        PH(8)
        const unsigned long long ballot = __ballot(!reject);
        const int slot = it % E.wcap;
        if (!reject) {                                    // oldpar = newpar (+ the history row): this lane's half
            double *hrow = E.hist ? E.hist + ((size_t)tile * E.wcap + slot) * (size_t)E.hs * 64 : nullptr;
            for (int k0 = kh0; k0 < kh1; k0 += CBH) {
                double v[CBH];
#pragma unroll
                for (int u = 0; u < CBH; ++u) v[u] = GV(cand_t, (k0 + u < kh1) ? k0 + u : kh1 - 1);
#pragma unroll
                for (int u = 0; u < CBH; ++u) if (k0 + u < kh1) { GV(theta_t, k0 + u) = v[u]; if (hrow) GV(hrow, k0 + u) = v[u]; }
            }
            if (hrow && h == 0) GV(hrow, d) = L.ss1;
        }
        if (E.hist || E.accmask) {                        // (uniform) the tile's 64-bit ballot from the two waves' halves
            if (pl == 0) sbal[w] = (uint32_t)ballot;
            if (E.record_s2 && h == 0) E.s2hist[((size_t)tile * E.wcap + slot) * 64 + lane] = L.sigma2;
            __syncthreads();
            if (tid == 0) {
                const unsigned long long m = (unsigned long long)sbal[0] | ((unsigned long long)sbal[1] << 32);
                if (E.hist) E.wacc[(size_t)tile * E.wcap + slot] = m;
                if (E.accmask) E.accmask[(size_t)(it - 1) * E.ntiles + tile] = m;
            }
        }
        PH(9)
    }
#ifdef MCX_PHASE_PROF
    if (pl == 0 && (tile == 0 || tile == E.ntiles / 2))
        printf("pooled_mfma2 tile %d wave %d its %d x10ns: normals %llu wait(a) %llu product %llu wait(b)+T+wait(c) %llu candidate %llu wait(d) %llu target+q %llu wait(e)+Q+wait(f) %llu sum+decide %llu accept %llu\n",
               tile, w, it1 - it0 + 1, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7], ph[8], ph[9]);
#endif
#undef PH
    if (h == 0) {
        lane_store(E, tile, lane, L);
        TIDX(E.ictr, tile, NICTR, I_ERSTAYED, lane) = erstayed;
    }
}

// ---------------------------------------------------------------- ... and with HALF A TILE PER WAVE (no workgroup at all)
// The two-wave workgroup above spends 47 % of its wave cycles parked at its six barriers.  The same split of the work WITHOUT a partner: one wave
// = 32 chains of a tile, two lanes per chain exactly as above, all (up to four) output blocks of the products for its own TWO chain groups (eight
// accumulators), an LDS vector of [d4][32] doubles -- 13.3 kB at npar 50, so twelve waves share a CU (three per SIMD at 168 registers) and no wave
// ever waits for another.  Per chain the arithmetic of pooled_mfma_kernel; the tile's ballot is written as its two 32-bit halves.
#ifndef MCX_POOLED3_KU
#define MCX_POOLED3_KU MCX_POOLED_KU
#endif
template <bool TRI>
MCX_DEV void mfma_product_half(const double *__restrict__ M, const double *X, int pl, int d, int d4, int nt, mcx_d4 (&c)[4][2])
{
    const int li = pl & 15, lk = pl >> 4;
#pragma unroll
    for (int b = 0; b < 4; ++b) { c[b][0] = mcx_d4{0.0, 0.0, 0.0, 0.0}; c[b][1] = mcx_d4{0.0, 0.0, 0.0, 0.0}; }
    int kmax = d4;
    if (TRI) { const int last = 16 * nt; kmax = last < d4 ? last : d4; }
    const double *__restrict__ ap = M + (size_t)lk * d + li;
    const double *xp = X + lk * 32 + li;
    constexpr int KU = MCX_POOLED3_KU;
    for (int s0 = 0; s0 < kmax; s0 += 4 * KU) {
        double a[KU][4];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int s = (s0 + 4 * u < kmax) ? s0 + 4 * u : kmax - 4;
#pragma unroll
            for (int b = 0; b < 4; ++b) a[u][b] = ap[(size_t)s * d + 16 * (b < nt ? b : 0)];
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int s = s0 + 4 * u;
            if (s < kmax) {
                const double *xq = xp + s * 32;
                const double b0 = xq[0], b1 = xq[16];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (b < nt && (!TRI || s < 16 * (b + 1))) {
                        c[b][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][b], b0, c[b][0], 0, 0, 0);
                        c[b][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][b], b1, c[b][1], 0, 0, 0);
                    }
                }
            }
        }
    }
}

#ifndef MCX_POOLED3_WAVES
#define MCX_POOLED3_WAVES 3
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MCX_POOLED3_WAVES, MCX_POOLED3_WAVES)))
void pooled_mfma3_kernel(EngineDev E, int it0, int it1, const double *__restrict__ g_mu, const double *__restrict__ g_lamT, const double *__restrict__ g_RT)
{
    extern __shared__ double X[];                        // the half tile's vector [d4][32]
    const int pl = threadIdx.x, w = blockIdx.x & 1, tile = blockIdx.x >> 1, d = E.d;
    const int xc = pl & 31, lane = 32 * w + xc, h = pl >> 5;      // xc: the chain's column of X; lane: its index in the tile (GV / TIDX)
    const int d4 = (d + 3) & ~3, nt = (d + 15) >> 4, li = pl & 15, lk = pl >> 4;
    const int kh0 = h ? (d + 1) / 2 : 0, kh1 = h ? d : (d + 1) / 2;
#define XH(k) X[(size_t)(k) * 32 + xc]
    double *theta_t = E.theta + (size_t)tile * d * 64;
    double *cand_t = E.cand + (size_t)tile * d * 64;
    const bool gauss = (E.tgt.kind == TGT_GAUSS);
    const bool plain = gauss && !E.tgt.lo && !E.tgt.hi && !E.tgt.pmu;
    LaneState L;
    lane_load(E, tile, lane, L);
    uint32_t erstayed = TIDX(E.ictr, tile, NICTR, I_ERSTAYED, lane);
    mcx_d4 c[4][2];
    constexpr int CBH = 16;
    for (int it = it0; it <= it1; ++it) {
        gen_normals2<MCX_POOLED2_NBL>(L.g, X + xc, 32, h, d, true);
        if (h == 0) for (int k = d; k < d4; ++k) XH(k) = 0.0;
        MCX_WAVE_LDS_SYNC();
        if (it == it1) {
            double *zk = E.zs + ((size_t)tile * 2 + (it & 1)) * d * 64;
            for (int k = kh0; k < kh1; ++k) GV(zk, k) = XH(k);
        }
        if (E.usesvd) mfma_product_half<false>(g_RT, X, pl, d, d4, nt, c);
        else mfma_product_half<true>(g_RT, X, pl, d, d4, nt, c);
        MCX_WAVE_LDS_SYNC();
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (b < nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * b + lk + 4 * r;
                    if (row < d4) { double *o = X + (size_t)row * 32 + li; o[0] = c[b][0][r]; o[16] = c[b][1][r]; }
                }
            }
        }
        MCX_WAVE_LDS_SYNC();
        for (int k0 = kh0; k0 < kh1; k0 += CBH) {
            double th[CBH], tv[CBH];
#pragma unroll
            for (int u = 0; u < CBH; ++u) { const int k = (k0 + u < kh1) ? k0 + u : kh1 - 1; th[u] = GV(theta_t, k); tv[u] = XH(k); }
#pragma unroll
            for (int u = 0; u < CBH; ++u) {
                if (k0 + u < kh1) {
                    const double cnd = th[u] + tv[u];
                    GV(cand_t, k0 + u) = cnd;
                    if (gauss) XH(k0 + u) = cnd - g_mu[k0 + u];
                }
            }
        }
        if (!plain) __threadfence_block();                // bounds / prior / a non-Gaussian target read the candidate's other half through global memory
        const bool inb = target_inbounds(E.tgt, d, lane, cand_t);
        const double pri2 = target_prior(E.tgt, d, lane, cand_t);
        double ss2;
        if (gauss) {
            if (h == 0) for (int k = d; k < d4; ++k) XH(k) = 0.0;
            MCX_WAVE_LDS_SYNC();
            mfma_product_half<false>(g_lamT, X, pl, d, d4, nt, c);      // y = Lam v
            double q[4][2];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int o0 = 16 * b + lk;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    double qq = c[b][g][0] * X[(size_t)(o0 < d4 ? o0 : 0) * 32 + 16 * g + li];
#pragma unroll
                    for (int r = 1; r < 4; ++r) {
                        const int o = o0 + 4 * r;
                        const double t = dfma(c[b][g][r], X[(size_t)(o < d4 ? o : 0) * 32 + 16 * g + li], qq);
                        qq = (o < d) ? t : qq;
                    }
                    q[b][g] = qq;
                }
            }
            MCX_WAVE_LDS_SYNC();
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (b < nt && 16 * b + lk < d) { X[(size_t)(4 * b + lk) * 32 + li] = q[b][0]; X[(size_t)(4 * b + lk) * 32 + 16 + li] = q[b][1]; }
            }
            MCX_WAVE_LDS_SYNC();
            ss2 = XH(0);
#pragma unroll 4
            for (int e = 1; e < 4 * nt; ++e) if (16 * (e >> 2) + (e & 3) < d) ss2 = ss2 + XH(e);
            MCX_WAVE_LDS_SYNC();                          // (the next iteration's normals overwrite these rows)
        } else {
            ss2 = target_ss<false>(E.tgt, d, lane, cand_t, g_mu, g_lamT);
        }
        bool reject;
        if (E.method == M_ER) {
            if (!inb) { L.bnd += 1; reject = true; }
            else {
                double u = rng_uniform(L.g);
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
        const unsigned long long ballot = __ballot(!reject);
        const int slot = it % E.wcap;
        if (!reject) {
            double *hrow = E.hist ? E.hist + ((size_t)tile * E.wcap + slot) * (size_t)E.hs * 64 : nullptr;
            for (int k0 = kh0; k0 < kh1; k0 += CBH) {
                double v[CBH];
#pragma unroll
                for (int u = 0; u < CBH; ++u) v[u] = GV(cand_t, (k0 + u < kh1) ? k0 + u : kh1 - 1);
#pragma unroll
                for (int u = 0; u < CBH; ++u) if (k0 + u < kh1) { GV(theta_t, k0 + u) = v[u]; if (hrow) GV(hrow, k0 + u) = v[u]; }
            }
            if (hrow && h == 0) GV(hrow, d) = L.ss1;
        }
        if (E.hist) {
            if (pl == 0) ((uint32_t *)&E.wacc[(size_t)tile * E.wcap + slot])[w] = (uint32_t)ballot;     // this half's 32 bits of the tile's ballot
            if (E.record_s2 && h == 0) E.s2hist[((size_t)tile * E.wcap + slot) * 64 + lane] = L.sigma2;
        }
        if (E.accmask && pl == 0) ((uint32_t *)&E.accmask[(size_t)(it - 1) * E.ntiles + tile])[w] = (uint32_t)ballot;
    }
#undef XH
    if (h == 0) {
        lane_store(E, tile, lane, L);
        TIDX(E.ictr, tile, NICTR, I_ERSTAYED, lane) = erstayed;
    }
}

} // namespace mcx
