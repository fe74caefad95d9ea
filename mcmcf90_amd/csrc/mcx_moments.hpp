// mcx_moments.hpp -- pooled moments of the current states (the one exchanged vector of the multi-GPU path), the fixed pairwise tree, debug
// probes (one of the family headers mcx_kernels.hpp includes, in this order: mcx_common, mcx_products, mcx_step, mcx_scam, mcx_pooled,
// mcx_phase, mcx_adapt, mcx_svd, mcx_moments)
#pragma once
#include "mcx_svd.hpp"

namespace mcx {

// The xor-butterfly's tree over 64 values (adjacent lanes first: v_l + v_{l^1}, (..) + (..)_{l^2}, ...; the sum it leaves in lane 0)
// for values that are PRODUCED as they are needed: eight at a time summed in their tree, the eight group sums through a binary counter
// of three partial sums -- the same additions with the same operands on the same sides as the array form, so the same bits.  The loop
// over the groups is a real loop: unrolled, the scheduler asks for all 128 LDS values first (the array form cost moments_kernel 260
// registers and all but one block per CU; a fully unrolled counter under __launch_bounds__(256, 4) spilled 20 of them).  Round 6.
template <class F>
MCX_DEV double tree64(F &&val)
{
    double s3 = 0.0, s4 = 0.0, s5 = 0.0, res = 0.0;      // sums of 8, 16, 32 values waiting for their partner
#pragma clang loop unroll(disable)
    for (int g = 0; g < 8; ++g) {
        const int l = 8 * g;
        const double v0 = val(l), v1 = val(l + 1), v2 = val(l + 2), v3 = val(l + 3);
        const double v4 = val(l + 4), v5 = val(l + 5), v6 = val(l + 6), v7 = val(l + 7);
        double v = ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7));
        if (!(g & 1)) { s3 = v; continue; }
        v = s3 + v;
        if (!(g & 2)) { s4 = v; continue; }
        v = s4 + v;
        if (!(g & 4)) { s5 = v; continue; }
        res = s5 + v;
    }
    return res;
}

// ---------------------------------------------------------------- pooled moments of the current states
// out[tile][1 + d + d(d+1)/2]: partial sums over the 64 lanes of a tile by an xor-butterfly (a fixed
// pairwise tree: adjacent lanes first); the host finishes the tree over tiles, RCCL over GPUs.
// Second moments are indexed j(j+1)/2 + i for i <= j.
// kind 0: [count, sum_j x_j, sum x_i x_j (i <= j)], x = theta - par0                       (1 + d + P terms)
// kind 1: the same followed by sum_c stayed_c (the pooled rejection count of a burn-in tick)  (2 + d + P)
// kind 2: the pooled RAM statistic of iteration `it` (MCMC_run_ram.F90:166-172 summed over chains): [count, sum alpha,
//         sum_c sign(a_c) x_c x_c'], x_c = u_c / sum(u_c**2) * a_c, a_c = rs (alpha_c - alphatarget)     (2 + P)
// BIG (npar >= 316: (64 (d | 1) + 320) doubles exceed 160 KiB -- the tile's 64 vectors no longer fit a CU's LDS): the same terms with every
// x value formed from global memory where it is used -- the same operations on the same operands, so the same bits; only the four
// 64-vectors (count, alpha or stayed, sign, sum(u**2)) and the chains' a = rs (alpha - alphatarget) stay in LDS.  Slower (each term reads
// its 2 x 64 values through L2); any npar.
template <bool BIG>
__global__ __launch_bounds__(256) void moments_kernel(EngineDev E, double *out, int nchains, int kind, int it, double rs)
{
    // The tile's 64 vectors x_c go to LDS once (chain-major, odd stride); then each of the 256 threads takes terms m, m + 256, ...:
    // it forms the term's 64 values (one per chain) and adds them in the butterfly's tree order (lane pairs first) -- the sums
    // v_l + v_{l^1}, (..) + (..)_{l^2}, ... a xor-butterfly leaves in lane 0 -- as they come (tree64).  No barrier after the first, every
    // lane on a term of its own; the old form (one chain per lane, 64 terms at a time transposed through LDS) spent its time
    // in the latencies of 1300 global loads and 2 x 20 barriers per tile at four waves per CU.
    extern __shared__ double XS[];                      // x[64][DP]; then count[64], alpha or stayed [64], sign(a) [64], sum(u**2) [64]
    const int tid = threadIdx.x, tile = blockIdx.x, d = E.d, P = E.P, DP = BIG ? 0 : (d | 1);
    double *sp0 = XS + (size_t)64 * DP, *sp1 = sp0 + 64, *sg = sp1 + 64, *ssu = sg + 64, *sa = ssu + 64;
    const double *theta_t = E.theta + (size_t)tile * d * 64;
    const double *z_t = E.zs + ((size_t)tile * 2 + (it & 1)) * d * 64;        // kind 2: the normals iteration `it` proposed with
    const int len = (kind == 2) ? 2 + P : (1 + d + P + (kind == 1 ? 1 : 0));
    double *o = out + (size_t)tile * len;
    const int c0 = tid & 63;
    const bool act = (tile * 64 + c0) < nchains;
    if (tid < 64) {
        sp0[c0] = act ? 1.0 : 0.0;
        double s1 = 0.0, sgn = 1.0, su = 1.0;
        if (kind == 2) {
            const double alpha = TIDX(E.scal, tile, NSCAL, S_ALPHA12, c0);
            const double a = rs * (alpha - E.alphatarget);
            su = 0.0;
            for (int k = 0; k < d; ++k) { const double z = GV2(z_t, k, c0); su = su + z * z; }
            s1 = act ? alpha : 0.0;
            sgn = (!act || a >= 0.0) ? 1.0 : -1.0;
        } else if (kind == 1) s1 = act ? (double)TIDX(E.ictr, tile, NICTR, I_STAYED, c0) : 0.0;
        sp1[c0] = s1; sg[c0] = sgn; ssu[c0] = su;
        if (BIG) sa[c0] = (kind == 2) ? rs * (TIDX(E.scal, tile, NSCAL, S_ALPHA12, c0) - E.alphatarget) : 0.0;
    }
    __syncthreads();
    if (!BIG) {
    if (kind == 2) {
        const double alpha = TIDX(E.scal, tile, NSCAL, S_ALPHA12, c0);
        const double a = rs * (alpha - E.alphatarget), su = ssu[c0];
        for (int k = tid >> 6; k < d; k += 4) XS[(size_t)c0 * DP + k] = act ? GV2(z_t, k, c0) / su * a : 0.0;     // x = u / sum(u**2) * a
    } else {
        for (int k = tid >> 6; k < d; k += 4) XS[(size_t)c0 * DP + k] = act ? (GV2(theta_t, k, c0) - E.par0[k]) : 0.0;
    }
    __syncthreads();
    }
    // x value k of chain l of the tile: from the LDS copy, or (BIG) formed here
    auto xv = [&](int l, int k) -> double {
        if (!BIG) return XS[(size_t)l * DP + k];
        if (!(sp0[l] != 0.0)) return 0.0;
        return (kind == 2) ? GV2(z_t, k, l) / ssu[l] * sa[l] : (GV2(theta_t, k, l) - E.par0[k]);
    };
    const int pair0 = (kind == 2) ? 2 : 1 + d;          // first second-moment term
    for (int m = tid; m < len; m += 256) {
        double r;
        if (m == 0) r = tree64([&](int l) { return sp0[l]; });
        else if (m < pair0 && kind == 2) r = tree64([&](int l) { return sp1[l]; });
        else if (m < pair0) r = tree64([&](int l) { return xv(l, m - 1); });
        else if (m >= pair0 + P) r = tree64([&](int l) { return sp1[l]; });          // kind 1: the rejection counts
        else {
            const int q = m - pair0;                    // = j (j + 1) / 2 + i, i <= j
            int j = (int)((sqrt(8.0 * (double)q + 1.0) - 1.0) * 0.5);
            while ((j + 1) * (j + 2) / 2 <= q) ++j;
            while (j * (j + 1) / 2 > q) --j;
            const int i2 = q - j * (j + 1) / 2;
            if (kind == 2) r = tree64([&](int l) { const double t = xv(l, i2) * xv(l, j); return (sg[l] >= 0.0) ? t : -t; });
            else r = tree64([&](int l) { return xv(l, i2) * xv(l, j); });
        }
        o[m] = r;
    }
}

// one double into device memory in stream order (the rank's stop flag behind its moment vector): a pageable hipMemcpyAsync of eight bytes
// makes the host wait for the stream on this runtime, which put a host round trip between two bench steps
__global__ void set_double_kernel(double *p, double v) { *p = v; }

// Finish the pooled sum over tiles in the same fixed pairwise tree (adjacent tiles first):
//     for s = 1, 2, 4, ...: for t = 0, 2s, 4s, ... with t + s < ntiles: v[t] += v[t + s]
// v[0..len) of tile 0 ends up holding the result.  Deterministic and independent of how tiles are later grouped onto
// GPUs, as long as every GPU owns a power-of-two aligned block.  One launch runs six levels of the tree: thread
// (group g, moment k) loads the 64 partial sums v[(64 g + i) stride], i < 64, adds them up in registers in tree order
// and stores the result where the tree leaves it, v[64 g stride]; the host repeats with stride 64, 4096, ... until one
// group is left.  Loads are coalesced along k and independent of each other.
__global__ __launch_bounds__(256) void moments_tree_kernel(double *v, int ntiles, int len, int stride, double *dst)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const long long t0 = (long long)blockIdx.y * 64 * stride;
    if (k >= len) return;
    double a[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const long long t = t0 + (long long)i * stride;
        a[i] = (t < ntiles) ? v[(size_t)t * len + k] : 0.0;
    }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1)
#pragma unroll
        for (int i = 0; i + s < 64; i += 2 * s)
            if (t0 + (long long)(i + s) * stride < ntiles) a[i] = a[i] + a[i + s];
    v[(size_t)t0 * len + k] = a[0];
    if (dst && gridDim.y == 1) dst[k] = a[0];
}

// ---------------------------------------------------------------- debug probes of the device primitives
// (tests/test_gpu_primitives.py compares them bit for bit with the oracle)
__global__ void debug_math_kernel(int op, int n, const double *a, const double *b, double *out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = a[i], y = b ? b[i] : 0.0, r = 0.0;
    switch (op) {
    case 0: r = d_log(x); break;
    case 1: r = d_exp(x); break;
    case 2: r = sqrt(x); break;
    case 3: r = x / y; break;
    case 4: r = dfma(x, y, x); break;
    case 5: { double c, s, rr; d_rotg(x, y, rr, c, s); r = rr + c * 3.0 + s * 7.0; break; }
    }
    out[i] = r;
}

// stream of one chain: kind 0 uniforms, 1 normals (normal_bm order), 2 gamma(a, b)
__global__ void debug_rng_kernel(uint32_t k0, uint32_t k1, int kind, int n, double a, double b, double *out, uint64_t *nused)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Rng g; g.k0 = k0; g.k1 = k1; g.n = 0; g.cblk = 0; g.c2 = g.c3 = 0; g.saved = 0; g.saved_y = 0.0;
    for (int i = 0; i < n; ++i) {
        if (kind == 0) out[i] = rng_uniform(g);
        else if (kind == 1) out[i] = rng_normal(g);
        else out[i] = rng_gamma(g, a, b);
    }
    *nused = g.n;
}

} // namespace mcx
