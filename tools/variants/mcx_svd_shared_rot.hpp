// tools/variants/mcx_svd_shared_rot.hpp -- a MEASURED NEGATIVE, not part of libmcmcx.so (tools/variants/README.md; build: tools/build_variant.sh NAME -DMCX_VARIANTS).
#pragma once
namespace mcx {

// svd_sweep_stream32_kernel with the rotations worked out ONCE per pair.  An octet's eight lanes hold eight partial chains of the three sums and, in
// the kernel above, all eight then run the same scalar tail -- the threshold's square root, zeta's division, the two square roots and the two
// divisions of t and c: six quarter-rate sequences, about half of a step's instructions -- four waves doing it for eight pairs each.  Here the
// octets leave (alpha, beta, gamma) in LDS, the first 32 lanes of wave 0 take one pair each, and everybody picks up (c, s) and a flag: the same
// operations on the same numbers, a quarter of the issue slots for the tail; two more workgroup barriers per step, which three workgroups per CU
// cover.  MEASURED NEGATIVE (round 5, the verdict's item 7): one adaptation of 16384 chains at npar 200 takes 0.688 s against 0.525 s with the kernel
// above -- a step is bound by the LATENCY of the tail's five dependent divide / square-root sequences (pinned), which this form lengthens by two
// barriers, not by its issue slots.  Kept selectable (MCMCX_SVD_SHARED_ROT=1) beside the other forms of the parity test; never the engine's choice.
template <int RL>
__global__ __launch_bounds__(256, MCX_SVDS_WAVES) void svd_sweep_stream32s_kernel(double *Gc, mcx_d2 *rot, uint8_t *state, int *any_rotated, int nlanes, int d)
{
    extern __shared__ double S[];
    __shared__ int s_rot;
    __shared__ double s_abg[3 * 32];
    __shared__ mcx_d2 s_cs[32];
    __shared__ int s_on[32];
    const int chain = blockIdx.x, tid = threadIdx.x;
    if (chain >= nlanes || state[chain] != 1) return;
    double *G = Gc + (size_t)chain * d * d;
    mcx_d2 *log = rot + (size_t)chain * ((size_t)d * (d - 1) / 2);
    constexpr int LS = 8 * RL + 2;
    double *GY = S;
    constexpr int b = 32, RB = b + 1;
    const int nb = (d + b - 1) / b;
    const int ol = tid >> 3, oj = tid & 7;
    const bool ld = tid < d;
    if (tid == 0) s_rot = 0;
    double xr[RL], yr[RL];
    double stg = 0.0;
    for (int e = tid; e < RB * LS; e += 256) GY[e] = 0.0;
    __syncthreads();
    for (int I = 0; I < nb; ++I) {
        const int I0 = I * b, wI = (d - I0) < b ? (d - I0) : b;
        const int nJ = d - I0 - 1;
        if (ol == 0) {
#pragma unroll
            for (int u = 0; u < RL; ++u) { const int k = oj + 8 * u; xr[u] = (k < d) ? G[(size_t)I0 * d + k] : 0.0; }
        }
        for (int e = tid; e < 2 * d; e += 256) {
            const int c = e / d, k = e - c * d;
            if (c < nJ) GY[(size_t)c * LS + k] = G[(size_t)(I0 + 1 + c) * d + k];
        }
        __syncthreads();
        const int nsteps = nJ + wI;
        for (int t = 0; t < nsteps; ++t) {
            {
                const int cs = t - wI, cw = t + 1, cg = t + 2;
                if (ld && cs >= wI - 1 && cs < nJ) G[(size_t)(I0 + 1 + cs) * d + tid] = GY[(size_t)(cs % RB) * LS + tid];
                if (ld && cw >= 2 && cw < nJ) GY[(size_t)(cw % RB) * LS + tid] = stg;
                if (ld && cg < nJ) stg = G[(size_t)(I0 + 1 + cg) * d + tid];
            }
            if (ol < wI && t == 2 * ol - 1) {
                const double *src = GY + (size_t)((ol - 1) % RB) * LS;
#pragma unroll
                for (int u = 0; u < RL; ++u) xr[u] = src[oj + 8 * u];
            }
            const int jj = t - ol;
            const bool pa = ol < wI && jj >= ol && jj < nJ;                  // this octet has a pair in this step
            double *ycol = GY + (size_t)((pa ? jj : 0) % RB) * LS;
            if (pa) {
#pragma unroll
                for (int u = 0; u < RL; ++u) yr[u] = ycol[oj + 8 * u];
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
                for (int u = 0; u < RL; ++u) { alpha = dfma(xr[u], xr[u], alpha); beta = dfma(yr[u], yr[u], beta); gamma = dfma(xr[u], yr[u], gamma); }
#pragma unroll
                for (int o = 1; o < 8; o <<= 1) {
                    alpha = alpha + __shfl_xor(alpha, o, 64); beta = beta + __shfl_xor(beta, o, 64); gamma = gamma + __shfl_xor(gamma, o, 64);
                }
                if (oj == 0) { s_abg[3 * ol] = alpha; s_abg[3 * ol + 1] = beta; s_abg[3 * ol + 2] = gamma; }
            }
            __syncthreads();
            if (tid < 32) {                                                 // pair-lane tid's rotation (or none)
                const int jq = t - tid;
                if (tid < wI && jq >= tid && jq < nJ) {
                    const double alpha = s_abg[3 * tid], beta = s_abg[3 * tid + 1], gamma = s_abg[3 * tid + 2];
                    mcx_d2 cs; cs.x = 1.0; cs.y = 0.0;
                    int on = 0;
                    if ((gamma != 0.0) && !(fabs(gamma) <= 1e-15 * sqrt(alpha * beta))) {
                        const double zeta = (beta - alpha) / (2.0 * gamma);
                        const double tt = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                        const double c = 1.0 / sqrt(1.0 + tt * tt);
                        cs.x = c; cs.y = c * tt;
                        on = 1; s_rot = 1;
                    }
                    s_cs[tid] = cs; s_on[tid] = on;
                    log[svd_pair_index(I0 + tid, I0 + 1 + jq, d)] = cs;
                }
            }
            __syncthreads();
            if (pa && s_on[ol]) {
                const mcx_d2 cs = s_cs[ol];
                const double c = cs.x, sn = cs.y;
#pragma unroll
                for (int u = 0; u < RL; ++u) {
                    const double a0 = xr[u], b0 = yr[u];
                    xr[u] = c * a0 - sn * b0; ycol[oj + 8 * u] = sn * a0 + c * b0;
                }
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

} // namespace mcx
