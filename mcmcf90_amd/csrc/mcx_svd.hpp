// mcx_svd.hpp -- the adaptation's SVD at large npar: blocked one-sided Jacobi (the pinned routine, oracle/mcx_svd.h), one workgroup per
// chain (one of the family headers mcx_kernels.hpp includes, in this order: mcx_common, mcx_products, mcx_step, mcx_scam, mcx_pooled,
// mcx_phase, mcx_adapt, mcx_svd, mcx_moments)
#pragma once
#include "mcx_adapt.hpp"

namespace mcx {

// ---------------------------------------------------------------- blocked one-sided Jacobi SVD, one workgroup per chain
// The pinned routine (oracle/mcx_svd.h; symsvd_dev in mcx_products.hpp runs it one lane per chain) streams four columns per pair from
// HBM: 640 kB of G and V per chain at npar = 200, ~10-26 sweeps x 19900 pairs.  A pair (p,q) only touches columns p and q, so any order
// of the pairs that keeps "(p,q) after (p,q-1) and after (p-1,q)" (and (p,p+1) after (p-1,p)) produces the same bits.  The kernels below
// use that freedom, one workgroup of 256 threads per chain:
//   * the three dot products of a pair are the routine's eight partial fma chains over the rows (rows j, j + 8, ...), one per lane of
//     an OCTET; the rows cannot be spread further -- that would change the summation order;
//   * pair-lane l of a column block I keeps the SAME column I0 + l for a whole pass -- only its partner changes -- so its octet holds
//     that column in registers, and a step is one phase: read the partner column from LDS, the three chains, the butterfly, the rotation
//     (all eight lanes derive it from the same operands), apply it, write the partner back;
//   * the partners of column I0 + l are simply the columns I0 + l + 1 .. npar - 1 in order: they are STREAMED past the block through a
//     ring of LDS columns (below);
//   * V is kept OUT of the sweep: it never feeds back into the rotations, so the sweep only logs (c, s) per pair and
//     svd_applyv_stream32_kernel replays the log on V afterwards, row-parallel and barrier-free (a wave owns its rows).
// One launch of each per sweep; the host stops when no chain rotated (mcx_api.hip: launch_adapt).  Storage is chain-major here (a chain's
// column = 8 d contiguous bytes); tile2chain_kernel / chain2tile_kernel convert from and to the engine's tile-interleaved layout.
// Earlier generations of these kernels (block pairs in LDS; the I block in registers without the stream; a shared scalar tail per pair)
// are bit-equal, slower, and no longer in the library: tools/variants/README.md.
// LDS column stride: even (16-byte accesses), = 2 mod 4 (16 lanes on 16 columns: 64 banks)
MCX_DEV int svd_ls(int d) { return ((d + 1) & ~1) + (((d + 1) & 2) ? 0 : 2); }
MCX_DEV size_t svd_pair_index(int p, int q, int d) { return (size_t)p * d - (size_t)p * (p + 1) / 2 + (size_t)(q - p - 1); }

// state[chain]: 0 = not part of this factorisation, 1 = sweeping, 2 = converged (its last sweep rotated nothing)
__global__ __launch_bounds__(256) void svd_init_kernel(double *Vc, uint8_t *state, const uint8_t *need, int nlanes, int d)
{
    const int chain = blockIdx.x, tid = threadIdx.x;
    if (chain >= nlanes) return;
    if (tid == 0) state[chain] = need[chain] ? 1 : 0;
    if (!need[chain]) return;
    double *V = Vc + (size_t)chain * d * d;
    for (int e = tid; e < d * d; e += 256) V[e] = (e % (d + 1) == 0) ? 1.0 : 0.0;
}


// The streamed sweep.  A block pair (I,J) worked on alone takes wI + wJ - 1 steps for wI wJ pairs: on average half of the pair-lanes have
// a partner.  Streaming removes that: pair-lane l meets stream column j (= column I0 + 1 + j) at step l + j, j >= l -- one wavefront over
// the whole rest of the matrix (the block's own columns are the stream's first wI - 1: pair-lane l takes column I0 + l out of the ring at
// step 2 l - 1, after its last pair as a partner), every pair-lane busy from its first partner to its last.  Two pairs that share a column
// keep their order, so do the bits.  A column is needed for wI consecutive steps: it enters a ring of wI + 2 LDS columns one step ahead and
// leaves it for global memory the step after its last pair.  This form (npar 201..256): each wave holds six octets and sixteen loader
// lanes; the loaders' global loads (issued one step before they write the ring) and stores run under the wave's own pair arithmetic.
#ifndef MCX_SVDS_WAVES
#define MCX_SVDS_WAVES 3                                     // waves per SIMD asked for up to npar 208 (RL 26)
#endif
#define MCX_SVDS_EPT 4                                       // elements of a column per loader lane: 64 loaders, npar <= 256
template <int RL>
__global__ __launch_bounds__(256, (RL <= 26 ? MCX_SVDS_WAVES : 2)) void svd_sweep_stream_kernel(double *Gc, mcx_d2 *rot, uint8_t *state,
    int *any_rotated, int nlanes, int d, int b)
{
    extern __shared__ double S[];
    __shared__ int s_rot;
    const int chain = blockIdx.x, tid = threadIdx.x;
    if (chain >= nlanes || state[chain] != 1) return;
    double *G = Gc + (size_t)chain * d * d;
    mcx_d2 *log = rot + (size_t)chain * ((size_t)d * (d - 1) / 2);
    // ring column stride: every octet row 8 u + oj exists (rows >= npar hold zeros: they add nothing to the three sums and rotate to zero
    // -- no bounds tests in the loop); = 2 mod 4
    constexpr int LS = 8 * RL + 2;
    double *GY = S;                                            // the ring: RB columns
    const int RB = b + 2;
    const int nb = (d + b - 1) / b;
    const int wv = tid >> 6, ln = tid & 63;
    const bool loader = ln >= 48;
    const int ol = loader ? 64 : wv * 6 + (ln >> 3), oj = ln & 7;   // pair-lane of this thread's octet, partial chain / row residue
    const int li = wv * 16 + (ln - 48);                        // loader lanes: 0 .. 63
    if (tid == 0) s_rot = 0;
    double xr[RL];
    double stg[MCX_SVDS_EPT];                                  // loaders: the column on its way from global memory to the ring
    auto pair_step = [&](double *ycol, size_t logidx) __attribute__((always_inline)) {
        double yr[RL];
#pragma unroll
        for (int u = 0; u < RL; ++u) yr[u] = ycol[oj + 8 * u];
        double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
        for (int u = 0; u < RL; ++u) { alpha = dfma(xr[u], xr[u], alpha); beta = dfma(yr[u], yr[u], beta); gamma = dfma(xr[u], yr[u],
            gamma); }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            alpha = alpha + __shfl_xor(alpha, o, 64); beta = beta + __shfl_xor(beta, o, 64); gamma = gamma + __shfl_xor(gamma, o, 64);
        }
        mcx_d2 cs; cs.x = 1.0; cs.y = 0.0;
        if ((gamma != 0.0) && !(fabs(gamma) <= 1e-15 * sqrt(alpha * beta))) {
            const double zeta = (beta - alpha) / (2.0 * gamma);
            const double tt = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            const double c = 1.0 / sqrt(1.0 + tt * tt);
            cs.x = c; cs.y = c * tt;
            const double sn = cs.y;
#pragma unroll
            for (int u = 0; u < RL; ++u) {
                const double a0 = xr[u], b0 = yr[u];
                xr[u] = c * a0 - sn * b0; ycol[oj + 8 * u] = sn * a0 + c * b0;
            }
            if (oj == 0) s_rot = 1;
        }
        if (oj == 0) log[logidx] = cs;
    };
    for (int e = tid; e < RB * LS; e += 256) GY[e] = 0.0;
    __syncthreads();
    for (int I = 0; I < nb; ++I) {
        const int I0 = I * b, wI = (d - I0) < b ? (d - I0) : b;
        const int nJ = d - I0 - 1;                             // stream columns: j = 0 .. nJ - 1 is column I0 + 1 + j
        if (ol == 0) {
#pragma unroll
            for (int u = 0; u < RL; ++u) { const int k = oj + 8 * u; xr[u] = (k < d) ? G[(size_t)I0 * d + k] : 0.0; }
        }
        for (int e = tid; e < 2 * d; e += 256) {                // the ring's first two columns
            const int c = e / d, k = e - c * d;
            if (c < nJ) GY[(size_t)c * LS + k] = G[(size_t)(I0 + 1 + c) * d + k];
        }
        __syncthreads();
        const int nsteps = nJ + wI;                            // (the last one only writes the last column back)
        for (int t = 0; t < nsteps; ++t) {
            if (loader) {
                const int cw = t + 1;                          // ring <- stream column cw (its load was issued in the previous step)
                if (cw >= 2 && cw < nJ) {
                    double *dst = GY + (size_t)(cw % RB) * LS;
#pragma unroll
                    for (int u = 0; u < MCX_SVDS_EPT; ++u) { const int k = li + 64 * u; if (k < d) dst[k] = stg[u]; }
                }
                const int cg = t + 2;                          // issue the load of stream column cg
                if (cg < nJ) {
                    const double *src = G + (size_t)(I0 + 1 + cg) * d;
#pragma unroll
                    for (int u = 0; u < MCX_SVDS_EPT; ++u) { const int k = li + 64 * u; if (k < d) stg[u] = src[k]; }
                }
                const int cs = t - wI;                         // stream column cs had its last pair in the previous step
                if (cs >= wI - 1 && cs < nJ) {
                    const double *src = GY + (size_t)(cs % RB) * LS;
                    double *dst = G + (size_t)(I0 + 1 + cs) * d;
#pragma unroll
                    for (int u = 0; u < MCX_SVDS_EPT; ++u) { const int k = li + 64 * u; if (k < d) dst[k] = src[k]; }
                }
            } else if (ol < wI) {
                // this pair-lane's own column: its last pair as a partner was in step 2 ol - 2
                if (t == 2 * ol - 1) {
                    const double *src = GY + (size_t)((ol - 1) % RB) * LS;
#pragma unroll
                    for (int u = 0; u < RL; ++u) xr[u] = src[oj + 8 * u];
                }
                const int jj = t - ol;
                if (jj >= ol && jj < nJ) pair_step(GY + (size_t)(jj % RB) * LS, svd_pair_index(I0 + ol, I0 + 1 + jj, d));
            }
            __syncthreads();
        }
        if (ol < wI) {
#pragma unroll
            for (int u = 0; u < RL; ++u) { const int k = oj + 8 * u; if (k < d) G[(size_t)(I0 + ol) * d + k] = xr[u]; }
        }
        __syncthreads();
    }
    if (tid == 0) { if (s_rot) *any_rotated = 1; else state[chain] = 2; }
}

// svd_sweep_stream_kernel with ALL lanes on pairs (npar <= 200): 32 pair-lanes, and every thread carries one element of the column
// entering the ring and of the one leaving it, so no lane is set aside for loading.  The ring is 33 columns of 8 RL + 2 <= 202 doubles
// (a leaving column hands its slot to the entering one element by element inside one thread): 53 kB, three workgroups per CU as before.
// 921 steps per sweep at npar 200 instead of 1134, 64 live lanes per wave instead of 48.
template <int RL>
__global__ __launch_bounds__(256, MCX_SVDS_WAVES) void svd_sweep_stream32_kernel(double *Gc, mcx_d2 *rot, uint8_t *state, int *any_rotated,
    int nlanes, int d)
{
    extern __shared__ double S[];
    __shared__ int s_rot;
    const int chain = blockIdx.x, tid = threadIdx.x;
    if (chain >= nlanes || state[chain] != 1) return;
    double *G = Gc + (size_t)chain * d * d;
    mcx_d2 *log = rot + (size_t)chain * ((size_t)d * (d - 1) / 2);
    // ring column stride: every octet row 8 u + oj exists (rows >= npar hold zeros: they add nothing to the three sums and rotate to zero
    // -- no bounds tests in the loop); = 2 mod 4
    constexpr int LS = 8 * RL + 2;
    double *GY = S;                                            // the ring: RB columns
    constexpr int b = 32, RB = b + 1;
    const int nb = (d + b - 1) / b;
    const int ol = tid >> 3, oj = tid & 7;                     // pair-lane of this thread's octet, partial chain / row residue
    // ... and every thread moves element `tid` of the columns on their way in and out
    const bool ld = tid < d;
    if (tid == 0) s_rot = 0;
    double xr[RL];
    double stg = 0.0;                                          // the element on its way from global memory to the ring
    auto pair_step = [&](double *ycol, size_t logidx) __attribute__((always_inline)) {
        double yr[RL];
#pragma unroll
        for (int u = 0; u < RL; ++u) yr[u] = ycol[oj + 8 * u];
        double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
        for (int u = 0; u < RL; ++u) { alpha = dfma(xr[u], xr[u], alpha); beta = dfma(yr[u], yr[u], beta); gamma = dfma(xr[u], yr[u],
            gamma); }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            alpha = alpha + __shfl_xor(alpha, o, 64); beta = beta + __shfl_xor(beta, o, 64); gamma = gamma + __shfl_xor(gamma, o, 64);
        }
        mcx_d2 cs; cs.x = 1.0; cs.y = 0.0;
        if ((gamma != 0.0) && !(fabs(gamma) <= 1e-15 * sqrt(alpha * beta))) {
            const double zeta = (beta - alpha) / (2.0 * gamma);
            const double tt = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            const double c = 1.0 / sqrt(1.0 + tt * tt);
            cs.x = c; cs.y = c * tt;
            const double sn = cs.y;
#pragma unroll
            for (int u = 0; u < RL; ++u) {
                const double a0 = xr[u], b0 = yr[u];
                xr[u] = c * a0 - sn * b0; ycol[oj + 8 * u] = sn * a0 + c * b0;
            }
            if (oj == 0) s_rot = 1;
        }
        if (oj == 0) log[logidx] = cs;
    };
    for (int e = tid; e < RB * LS; e += 256) GY[e] = 0.0;
    __syncthreads();
    for (int I = 0; I < nb; ++I) {
        const int I0 = I * b, wI = (d - I0) < b ? (d - I0) : b;
        const int nJ = d - I0 - 1;                             // stream columns: j = 0 .. nJ - 1 is column I0 + 1 + j
        if (ol == 0) {
#pragma unroll
            for (int u = 0; u < RL; ++u) { const int k = oj + 8 * u; xr[u] = (k < d) ? G[(size_t)I0 * d + k] : 0.0; }
        }
        for (int e = tid; e < 2 * d; e += 256) {                // the ring's first two columns
            const int c = e / d, k = e - c * d;
            if (c < nJ) GY[(size_t)c * LS + k] = G[(size_t)(I0 + 1 + c) * d + k];
        }
        __syncthreads();
        const int nsteps = nJ + wI;                            // (the last one only writes the last column back)
        for (int t = 0; t < nsteps; ++t) {
            {
                // ring slot (t + 1) mod RB changes hands: stream column t - wI (its last pair was in the previous step) leaves it for
                // global memory and column t + 1 (loaded in the previous step) enters -- element by element in the same thread, so RB = wI
                // + 1 will do
                const int cs = t - wI, cw = t + 1, cg = t + 2;
                if (ld && cs >= wI - 1 && cs < nJ) G[(size_t)(I0 + 1 + cs) * d + tid] = GY[(size_t)(cs % RB) * LS + tid];
                if (ld && cw >= 2 && cw < nJ) GY[(size_t)(cw % RB) * LS + tid] = stg;
                if (ld && cg < nJ) stg = G[(size_t)(I0 + 1 + cg) * d + tid];
            }
            if (ol < wI) {
                // this pair-lane's own column: its last pair as a partner was in step 2 ol - 2
                if (t == 2 * ol - 1) {
                    const double *src = GY + (size_t)((ol - 1) % RB) * LS;
#pragma unroll
                    for (int u = 0; u < RL; ++u) xr[u] = src[oj + 8 * u];
                }
                const int jj = t - ol;
                if (jj >= ol && jj < nJ) pair_step(GY + (size_t)(jj % RB) * LS, svd_pair_index(I0 + ol, I0 + 1 + jj, d));
            }
            __syncthreads();
        }
        if (ol < wI) {
#pragma unroll
            for (int u = 0; u < RL; ++u) { const int k = oj + 8 * u; if (k < d) G[(size_t)(I0 + ol) * d + k] = xr[u]; }
        }
        __syncthreads();
    }
    if (tid == 0) { if (s_rot) *any_rotated = 1; else state[chain] = 2; }
}


// Replays one sweep's rotations on V in the sweep's order.  The rotations touch the rows of V independently, so each of a chain's four
// waves (two row groups of 32 lanes) is a workgroup of its own with a ring of 33 columns of ITS rows in LDS, and never waits for the
// others: within a wave the LDS accesses of consecutive steps are ordered, no barrier is needed.  Thread (rl, rk0) keeps its rows of
// column I0 + rl in registers for the whole pass; only the partner column goes through the ring; lane rl < RP also carries row pair rl of
// the column entering the ring and of the one leaving it.  A pair-lane reads the next step's (c, s) from the log one step ahead.  Chains
// whose sweep rotated nothing (state 2) are skipped.  blockIdx: the four waves of a chain on one XCD (they read the same log).
template <int RP>         // row PAIRS per thread: 16 RP >= npar
__global__ __launch_bounds__(64) void svd_applyv_stream32_kernel(double *Vc, const mcx_d2 *rot, const uint8_t *state, int nlanes, int d)
{
    extern __shared__ double S[];
    const int blk = blockIdx.x;
    const int chain = (blk >> 5) * 8 + (blk & 7), wv = (blk >> 3) & 3;
    if (chain >= nlanes || state[chain] != 1) return;
    double *V = Vc + (size_t)chain * d * d;
    const mcx_d2 *log = rot + (size_t)chain * ((size_t)d * (d - 1) / 2);
    // doubles per row group (its odd last row at 2 RP) and per ring column (SLOT / 2 odd: 16 lanes on 16 columns, 64 banks)
    constexpr int RGS = 2 * RP + 2, SLOT = 4 * RP + 6;
    constexpr int b = 32, RB = b + 1;
    const int nb = (d + b - 1) / b;
    const int ln = threadIdx.x, rg = ln >> 5, rl = ln & 31, rk0 = 2 * wv + rg;
    const int lu = rl;                                         // ... and row pair `rl` (rl < RP) of the columns on their way in and out
    const int lk_ = 2 * rk0 + 16 * lu;
    const bool ld = lu < RP && lk_ + 1 < d;
    const bool oddrow = (d & 1) && rk0 == 0;                    // row d - 1 of an odd npar: row group 0's extra element
    double *ring = S + rg * RGS;
    mcx_d2 vr[RP];
    double vlast = 0.0;
    double stg[2] = {0.0, 0.0}, stgl = 0.0;                     // the row pair on its way to the ring (+ the odd row: lane rl = RP)
    auto g_load = [&](const double *col) __attribute__((always_inline)) {
        if (ld) { stg[0] = col[lk_]; stg[1] = col[lk_ + 1]; }
        if (oddrow && lu == RP) stgl = col[d - 1];
    };
    auto r_write = [&](double *slot) __attribute__((always_inline)) {
        if (ld) { mcx_d2 v2; v2.x = stg[0]; v2.y = stg[1]; *(mcx_d2 *)(slot + 2 * lu) = v2; }
        if (oddrow && lu == RP) slot[2 * RP] = stgl;
    };
    auto r_store = [&](const double *slot, double *col) __attribute__((always_inline)) {
        if (ld) { const mcx_d2 v2 = *(const mcx_d2 *)(slot + 2 * lu); col[lk_] = v2.x; col[lk_ + 1] = v2.y; }
        if (oddrow && lu == RP) col[d - 1] = slot[2 * RP];
    };
    auto pair_step = [&](double *vq, const mcx_d2 cs) __attribute__((always_inline)) {
        if (cs.x == 1.0 && cs.y == 0.0) return;
        const double c = cs.x, sn = cs.y;
        mcx_d2 vb[RP];
#pragma unroll
        // (row pairs beyond npar: zeros in the ring and in vr, they stay zero)
        for (int u = 0; u < RP; ++u) vb[u] = *(mcx_d2 *)(vq + 2 * u);
#pragma unroll
        for (int u = 0; u < RP; ++u) {
            mcx_d2 nva, nvb;
            nva.x = c * vr[u].x - sn * vb[u].x; nva.y = c * vr[u].y - sn * vb[u].y; nvb.x = sn * vr[u].x + c * vb[u].x;
                nvb.y = sn * vr[u].y + c * vb[u].y;
            vr[u] = nva; *(mcx_d2 *)(vq + 2 * u) = nvb;
        }
        if (oddrow) { const double va0 = vlast, vb0 = vq[2 * RP]; vlast = c * va0 - sn * vb0; vq[2 * RP] = sn * va0 + c * vb0; }
    };
    for (int e = ln; e < RB * SLOT; e += 64) S[e] = 0.0;
    __syncthreads();
    for (int I = 0; I < nb; ++I) {
        const int I0 = I * b, wI = (d - I0) < b ? (d - I0) : b;
        const int nJ = d - I0 - 1;
        if (rl == 0) {                                          // the block's first column: straight into registers
            const double *col = V + (size_t)I0 * d;
#pragma unroll
            for (int u = 0; u < RP; ++u) { const int k = 2 * rk0 + 16 * u; vr[u].x = 0.0; vr[u].y = 0.0; if (k + 1 < d) { vr[u].x = col[k];
                vr[u].y = col[k + 1]; } }
            if (oddrow) vlast = col[d - 1];
        }
        for (int c = 0; c < 2 && c < nJ; ++c) { g_load(V + (size_t)(I0 + 1 + c) * d); r_write(ring + (size_t)c * SLOT); }
        // log entry of this pair-lane's first pair (step 2 rl), the next ones follow it
        const size_t base = svd_pair_index(I0 + rl, I0 + 1 + rl, d);
        mcx_d2 nxt; nxt.x = 1.0; nxt.y = 0.0;
        if (rl == 0 && nJ > 0) nxt = log[base];
        const int nsteps = nJ + wI;
        for (int t = 0; t < nsteps; ++t) {
            // a step reads what other lanes of THIS wave wrote in the previous one: keep the compiler from moving LDS accesses across the
            // step boundary (the hardware runs a wave's LDS operations in order)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            {
                // slot (t + 1) mod RB changes hands, element by element in the lane that moves it: column t - wI out, column t + 1 in
                const int cs = t - wI, cw = t + 1, cg = t + 2;
                if (cs >= wI - 1 && cs < nJ) r_store(ring + (size_t)(cs % RB) * SLOT, V + (size_t)(I0 + 1 + cs) * d);
                if (cw >= 2 && cw < nJ) r_write(ring + (size_t)(cw % RB) * SLOT);
                if (cg < nJ) g_load(V + (size_t)(I0 + 1 + cg) * d);
            }
            if (rl < wI) {
                const mcx_d2 cur = nxt;
                const int jn = t + 1 - rl;                      // the next step's partner
                if (jn >= rl && jn < nJ) nxt = log[base + (size_t)(jn - rl)];
                if (t == 2 * rl - 1) {
                    const double *src = ring + (size_t)((rl - 1) % RB) * SLOT;
#pragma unroll
                    for (int u = 0; u < RP; ++u) vr[u] = *(const mcx_d2 *)(src + 2 * u);
                    if (oddrow) vlast = src[2 * RP];
                }
                const int jj = t - rl;
                if (jj >= rl && jj < nJ) pair_step(ring + (size_t)(jj % RB) * SLOT, cur);
            }
        }
        if (rl < wI) {
            double *col = V + (size_t)(I0 + rl) * d;
#pragma unroll
            for (int u = 0; u < RP; ++u) { const int k = 2 * rk0 + 16 * u; if (k + 1 < d) { col[k] = vr[u].x; col[k + 1] = vr[u].y; } }
            if (oddrow) col[d - 1] = vlast;
        }
        __syncthreads();                                       // (one wave: the next block row's loads follow these stores)
    }
}

// singular values = column norms of G (the routine's eight partial chains over the rows), sorted descending (first maximum wins), V's
// columns with them; the sorted vectors are left in G's place
__global__ __launch_bounds__(256) void svd_finish_kernel(double *Gc, const double *Vc, double *svc, const uint8_t *state, int nlanes, int d)
{
    __shared__ int s_perm[256];
    __shared__ double s_sv[256];
    const int chain = blockIdx.x, tid = threadIdx.x;
    if (chain >= nlanes || state[chain] == 0) return;
    double *G = Gc + (size_t)chain * d * d;
    const double *V = Vc + (size_t)chain * d * d;
    if (tid < d) {
        const double *gj = G + (size_t)tid * d;
        double pa[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) pa[u] = 0.0;
        for (int k0 = 0; k0 < d; k0 += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) if (k0 + u < d) pa[u] = dfma(gj[k0 + u], gj[k0 + u], pa[u]);
        }
        s_sv[tid] = sqrt(svd_tree8(pa)); s_perm[tid] = tid;
    }
    __syncthreads();
    if (tid == 0)
        for (int i = 0; i < d - 1; ++i) {
            int m = i; double sm = s_sv[i];
            for (int j = i + 1; j < d; ++j) if (s_sv[j] > sm) { m = j; sm = s_sv[j]; }
            if (m != i) { double ts = s_sv[i]; s_sv[i] = s_sv[m]; s_sv[m] = ts; int tp = s_perm[i]; s_perm[i] = s_perm[m]; s_perm[m] = tp; }
        }
    __syncthreads();
    if (tid < d) svc[(size_t)chain * d + tid] = s_sv[tid];
    for (int e = tid; e < d * d; e += 256) { const int j = e / d, k = e - j * d; G[e] = V[(size_t)s_perm[j] * d + k]; }
}

// tile-interleaved [tile][K][64 lanes]  <->  chain-major [chain][K], 64 x 64 blocks through LDS (both sides coalesced)
__global__ __launch_bounds__(256) void tile2chain_kernel(const double *__restrict__ src, double *__restrict__ dst, size_t K, size_t Kt,
    const uint8_t *need)
{
    __shared__ double T[64][65];
    const int tile = blockIdx.y, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const size_t k0 = (size_t)blockIdx.x * 64;
    bool any = false;
    for (int l = 0; l < 64; ++l) any = any || need[tile * 64 + l];
    if (!any) return;
    // element k0+r, lane tx (Kt: elements per tile on the interleaved side)
    for (int r = ty; r < 64; r += 4) if (k0 + r < K) T[r][tx] = src[((size_t)tile * Kt + k0 + r) * 64 + tx];
    __syncthreads();
    for (int c = ty; c < 64; c += 4) if (k0 + tx < K && need[tile * 64 + c]) dst[((size_t)tile * 64 + c) * K + k0 + tx] = T[tx][c];
}
__global__ __launch_bounds__(256) void chain2tile_kernel(const double *__restrict__ src, double *__restrict__ dst, size_t K, size_t Kt,
    const uint8_t *need)
{
    __shared__ double T[64][65];
    const int tile = blockIdx.y, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const size_t k0 = (size_t)blockIdx.x * 64;
    bool any = false;
    for (int l = 0; l < 64; ++l) any = any || need[tile * 64 + l];
    if (!any) return;
    for (int c = ty; c < 64; c += 4) if (k0 + tx < K && need[tile * 64 + c]) T[tx][c] = src[((size_t)tile * 64 + c) * K + k0 + tx];
    __syncthreads();
    for (int r = ty; r < 64; r += 4) if (k0 + r < K && need[tile * 64 + tx]) dst[((size_t)tile * Kt + k0 + r) * 64 + tx] = T[r][tx];
}

// one chain's lane of a tile-interleaved array: out[e] = src[e*64 + lane], e < n (mcmcx_get_chain copies a single
// chain's history to the host, not the other 63 of its tile)
__global__ __launch_bounds__(256) void gather_lane_kernel(const double *__restrict__ src, double *out, size_t n, int lane)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) out[e] = src[e * 64 + lane];
}

} // namespace mcx
