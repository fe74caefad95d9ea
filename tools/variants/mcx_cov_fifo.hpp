// tools/variants/mcx_cov_fifo.hpp -- a MEASURED NEGATIVE (round 6), not part of libmcmcx.so (tools/variants/README.md; -DMCX_VARIANTS, MCMCX_COV_FIFO=1).
// covmat_window_td (mcx_adapt.hpp) with the window's walk and the folds decoupled by a per-lane FIFO in LDS: a wave folds only where a lane's
// own chain accepted, so at 25 % accepted a lockstep walk runs ~100 masked folds for the ~30 each lane needs, and the FIFO form ~40 rounds in
// which nearly every lane works (same per-lane order, same operands: bit-equal on the whole suite).  Measured at the bench's regime -- the
// first thousand iterations from a narrow start, where 60-90 % of the proposals are accepted and the rounds are hardly fewer than the
// iterations -- it LOSES: the off-diagonal kernel 104.9 against 33.7 ms at config 4 --method dram, 3.08 against 0.94 ms at config 3 (the
// row loads that the lockstep form hides behind the fold before them are exposed in front of the LDS store, and the off-diagonal block's
// hundred accumulators leave one wave per SIMD to wait them out; its fold pays 650 accvgpr moves per round on top), the diagonal kernel
// 10.3 against 9.5 ms.  profiles/r06_c, profiles/r06_d.
#pragma once
namespace mcx {

// rows a lane can have waiting between the window's walk and its folds (LDS: CQ x (1 + TD | 2 TD) x 512 bytes per wave)
constexpr int CQ = 3;
template <bool DIAG>
MCX_DEV void covmat_window_td_fifo(const EngineDev &E, int it, int mode, int nblk)
{
    const int lane = threadIdx.x, d = E.d, P = E.P;
    const int w = blockIdx.x, j = w >> 3;
    const int tile = (j / nblk) * 8 + (w & 7);
    int blk = j % nblk;
    if (tile >= E.ntiles) return;
    const uint32_t flags = TIDX(E.ictr, tile, NICTR, I_ADFLAGS, lane);
    const bool act = (flags & ADF_STEADY) != 0;
    if (!__any(act)) return;
    const int nb = (d + TD - 1) / TD;
    int a0 = 0, b0 = 0;
    if (DIAG) { a0 = b0 = blk * TD; }
    // block row ar holds nb - 1 - ar off-diagonal blocks
    else { int ar = 0; while (blk >= nb - 1 - ar) { blk -= nb - 1 - ar; ++ar; } a0 = ar * TD; b0 = (ar + 1 + blk) * TD; }
    double *Ct = E.cmat + (size_t)tile * P * 64;
    const double *mean_t = E.mean + (size_t)tile * d * 64;
    const double *base_t = E.basetheta + (size_t)tile * d * 64;
    double *mnew_t = E.cand + (size_t)tile * d * 64;
    const bool unit = (mode & AD_BURN) != 0;                          // greedy restart: rows 1..it, unit weights, no base row
    const uint32_t count0 = unit ? 0u : TIDX(E.ictr, tile, NICTR, I_BASECNT, lane), adj0 = unit ? 0u : TIDX(E.ictr, tile, NICTR,
        I_LASTFREQ, lane);
    const int t0lane = unit ? 1 : (int)TIDX(E.ictr, tile, NICTR, I_WINSTART, lane), t1 = it;
    const double wsum = TIDX(E.scal, tile, NSCAL, S_WSUM, lane);
    int t0 = act ? t0lane : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int other = __shfl_xor(t0, o); t0 = other < t0 ? other : t0; }
    if (t0 == 0x7fffffff) return;
    const double *hist_t = E.hist + (size_t)tile * E.wcap * (size_t)E.hs * 64;
    const unsigned long long *wacc_t = (const unsigned long long *)E.wacc + (size_t)tile * E.wcap;
    constexpr int NBV = DIAG ? 1 : TD;                 // the b-side vectors exist for off-diagonal blocks only
    double C[TD][TD], ma[TD], xa[TD], mb[NBV], xb[NBV];
#pragma unroll
    for (int u = 0; u < TD; ++u) {
        const int a = (a0 + u < d) ? a0 + u : d - 1;
        ma[u] = GV(mean_t, a);
        xa[u] = unit ? 0.0 : GV(base_t, a);
        if (!DIAG) { const int b = (b0 + u < d) ? b0 + u : d - 1; mb[u] = GV(mean_t, b); xb[u] = unit ? 0.0 : GV(base_t, b); }
#pragma unroll
        for (int v = (DIAG ? u : 0); v < TD; ++v) {
            int bb = (b0 + v < d) ? b0 + v : d - 1;
            bb = bb < a ? a : bb;
            C[u][v] = GV(Ct, pidx(a, bb, d));
        }
    }
    double W = wsum;
    auto fold = [&](bool on, double w3) {
        if (on) {
            const double f1 = w3 / (W + w3 - 1.0), f2 = W / (W + w3), f3 = w3 / (W + w3);
#pragma unroll
            for (int u = 0; u < TD; ++u) { xa[u] = xa[u] - ma[u]; if (!DIAG) xb[u] = xb[u] - mb[u]; }
#pragma unroll
            for (int u = 0; u < TD; ++u) {
#pragma unroll
                for (int v = (DIAG ? u : 0); v < TD; ++v) {
                    double o = xa[u] * (DIAG ? xa[v] : xb[v]);
                    C[u][v] = C[u][v] + f1 * (f2 * o - C[u][v]);
                }
                if (!DIAG) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < TD; ++u) { ma[u] = ma[u] + f3 * xa[u]; if (!DIAG) mb[u] = mb[u] + f3 * xb[u]; }
            W = w3 + W;
        }
    };
    // The window is walked in lockstep (a row's loads are whole 512-byte segments whichever lanes want them), but a lane FOLDS only where
    // its
    // own chain accepted -- a quarter to a half of the iterations -- and a fold is five hundred operations under the exec mask of whoever
    // accepted at that iteration: up to round 5 a wave ran ~one fold per iteration of the window for the 25-45 each lane needs.  Round 6:
    // the walk (producer) and the folds (consumer) are decoupled by a per-lane FIFO of CQ rows in LDS.  The producer closes the lane's
    // previous row -- its weight is known now -- and appends (that weight, the new row); the consumer pops one entry PER LANE per round,
    // whatever iteration it came from, so nearly every lane works in every round and a window takes about as many rounds as its busiest
    // lane has rows (tools: 100 -> ~40 at 20 % accepted, ~54 at 30 %, ~70 at 45 %).  A lane's folds keep their order and their operands:
    // the same bits.  A round is forced when an accepting lane finds its FIFO full; the rest drains at the end.
    // [CQ][NQ][64]: entry = (weight of the row it closes | -1: none), the new row's NQ - 1 values
    extern __shared__ double cq[];
    constexpr int NQ = 1 + (DIAG ? TD : 2 * TD);
    int qn = 0, qhead = 0, qtail = 0;                    // entries waiting, next to pop, next to fill (slots modulo CQ)
    // producer: the lane has an open row; its last entry closes the window (no new row)
    bool have = act && !unit, term = false;
    uint32_t cnt = count0, adj = adj0;
    auto round = [&]() {                                 // every lane with an entry takes its oldest one
        const bool on = qn > 0;
        const bool last = on && term && qn == 1;         // the closing entry: a weight, no row
        const double *e = cq + ((size_t)(qhead % CQ) * NQ) * 64 + lane;
        const double w3 = on ? e[0] : -1.0;
        const bool fl = on && w3 >= 0.0;
        if (__any(fl)) fold(fl, w3);
        if (on) {
            if (!last) {
#pragma unroll
                for (int u = 0; u < TD; ++u) { xa[u] = e[(size_t)(1 + u) * 64]; if (!DIAG) xb[u] = e[(size_t)(1 + TD + u) * 64]; }
            }
            ++qhead; --qn;
        }
    };
    // ONE flat driver loop with ONE call of round().  What the register allocator made of the other shapes
    // (profiles/r06_c/fifo_shapes.txt):
    // round() inlined at three places (walk, close, drain): 1146 registers spilled, the kernel 23 x slower; one call inside a while-in-for
    // nest:
    // 358 spilled; this loop: none in the diagonal kernel (201 registers), two in the off-diagonal one -- whose fold needs a scheduling
    // barrier
    // per block row, or the scheduler, given a one-wave budget of 512 registers, hoists all hundred products and spills 188.
    // Each trip: the producer takes iteration t unless an accepting lane's FIFO is full; past the window it closes the open rows once there
    // is room; then it drains.  A trip that could not go on ends in a round.
    int t = t0;
    unsigned long long mine = 0ull;
    bool closed = false;
    for (;;) {
        bool want = false;                               // a round in this trip
        if (t <= t1) {
            const int q = (t - t0) & 63;
            if (q == 0) { const int tl = t + lane; mine = (tl <= t1) ? wacc_t[tl % E.wcap] : 0ull; }
            const unsigned long long m = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(mine >> 32), q) << 32)
                                         | (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)mine, q);
            const bool inwin = act && (t >= t0lane);
            const bool acc = inwin && ((m >> lane) & 1ull);
            want = __any(acc && qn == CQ);
            if (!want) {
                if (acc) {
                    const size_t so = (size_t)(t % E.wcap) * (size_t)E.hs * 64;
                    double *e = cq + ((size_t)(qtail % CQ) * NQ) * 64 + lane;
                    e[0] = have ? (unit ? 1.0 : (double)(cnt - adj)) : -1.0;
#pragma unroll
                    for (int u = 0; u < TD; ++u) {
                        e[(size_t)(1 + u) * 64] = hist_t[so + (size_t)((a0 + u < d) ? a0 + u : d - 1) * 64 + lane];
                        if (!DIAG) e[(size_t)(1 + TD + u) * 64] = hist_t[so + (size_t)((b0 + u < d) ? b0 + u : d - 1) * 64 + lane];
                    }
                    ++qtail; ++qn;
                    if (have) adj = 0;
                    have = true; cnt = 1;
                }
                if (inwin && !acc) cnt += 1;
                ++t;
            }
        } else if (!closed) {
            want = __any(have && qn == CQ);
            if (!want) {
                if (have) { cq[((size_t)(qtail % CQ) * NQ) * 64 + lane] = unit ? 1.0 : (double)(cnt - adj); ++qtail; ++qn; term = true; }
                closed = true;
            }
        } else {
            want = __any(qn > 0);
            if (!want) break;
        }
        if (want) round();
    }
#pragma unroll
    for (int u = 0; u < TD; ++u) {
        const int a = a0 + u;
#pragma unroll
        for (int v = (DIAG ? u : 0); v < TD; ++v) { const int b = b0 + v; if (act && a < d && b < d) GV(Ct, pidx(a, b, d)) = C[u][v]; }
        if (DIAG && act && a < d) GV(mnew_t, a) = ma[u];     // the other blocks still need the old means
    }
    if (DIAG && a0 == 0 && act) TIDX(E.scal, tile, NSCAL, S_WNEW, lane) = W;
}

__global__ __launch_bounds__(64, 2) void adapt_cov_diag_fifo_kernel(EngineDev E, int it, int mode, int nblk) { covmat_window_td_fifo<true>(E, it, mode, nblk); }
__global__ __launch_bounds__(64, 1) void adapt_cov_off_fifo_kernel(EngineDev E, int it, int mode, int nblk) { covmat_window_td_fifo<false>(E, it, mode, nblk); }

} // namespace mcx
