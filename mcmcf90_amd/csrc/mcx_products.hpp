// mcx_products.hpp -- the matrix-vector products of MCMC_propose (MCMC_DRAM.F90:20-31) on a lane's own factor: dtrmv('U','T') on the packed
// triangle in column panels, the full-matrix forms of the SVD paths, the lane-per-chain Jacobi SVD, the shared-table (pooled) forms
// (one of the family headers mcx_kernels.hpp includes, in this order: mcx_common, mcx_products, mcx_step, mcx_scam, mcx_pooled, mcx_phase,
// mcx_adapt, mcx_svd, mcx_moments)
#pragma once
#include "mcx_common.hpp"

namespace mcx {

// ---------------------------------------------------------------- proposal: P = R'z  (MCMC_DRAM.F90:20-31)
// dtrmv('U','T','N') (matutils.F90:108-109): p_j = sum_{i<=j} R(i,j) z_i, each dot product ascending in i
// as one fma chain from 0.  Column panels of PW accumulators in registers; every row contributes one
// contiguous PW x 512-byte segment, so the factor is read exactly once.
// PIPE: the AM / DRAM step kernels, where this product is the iteration's only pass over the factor and memory latency is
// what it waits for: three rows' loads in flight above the diagonal block, two inside it.  The RAM kernel runs it once per
// launch (later proposals come fused out of the update sweep) and keeps the plain form: its registers are spoken for.
template <bool PIPE>
MCX_DEV void trmv_panels(const double *Rt, const double *z_t, double *P_t, const double *theta_t, int lane, int d, bool act,
                         bool desc = false)
{
    const bool asc = act && !desc, dsc = act && desc;
    if (__any(asc)) {
        for (int J0 = 0; J0 < d; J0 += TW) {
            const int nw = (d - J0) < TW ? (d - J0) : TW;
            double P[TW];
#pragma unroll
            for (int u = 0; u < TW; ++u) P[u] = 0.0;
            if (asc) {
                // rows above the diagonal block, two rows' loads in flight: left to itself the compiler keeps four loads
                // outstanding (it sinks each load next to its fma), and a lane-per-chain wave then waits out the HBM
                // latency once per row; the accumulation order of every P[u] is unchanged (rows ascending)
                constexpr int NB = PIPE ? 3 : 1;                               // rows in flight
                double rr[NB][TW], zz[NB];
#define MCX_TRMV_LD(rv, zv, i_) { zv = GV(z_t, (i_)); const double *seg_ = Rt + (size_t)(rowstart((i_), d) + J0 - (i_)) * 64; \
                                  _Pragma("unroll") for (int u = 0; u < TW; ++u) rv[u] = LDNT(seg_, u < nw ? u : nw - 1); }
#define MCX_TRMV_FM(rv, zv) { _Pragma("unroll") for (int u = 0; u < TW; ++u) P[u] = dfma(rv[u], zv, P[u]); }
#pragma unroll
                for (int s_ = 0; s_ < NB - 1; ++s_) if (s_ < J0) MCX_TRMV_LD(rr[s_], zz[s_], s_)
                for (int i = 0; i < J0; i += NB) {
#pragma unroll
                    for (int s_ = 0; s_ < NB; ++s_) {
                        if (i + s_ + NB - 1 < J0) MCX_TRMV_LD(rr[(s_ + NB - 1) % NB], zz[(s_ + NB - 1) % NB], i + s_ + NB - 1)
                        if (i + s_ < J0) MCX_TRMV_FM(rr[s_], zz[s_])
                    }
                }
#undef MCX_TRMV_LD
#undef MCX_TRMV_FM
                // diagonal block: elements u >= ui; the next row's loads in flight
                {
                    double da[TW], db[TW], za = 0.0, zb = 0.0;
#define MCX_TRMV_LDD(rv, zv, i_) { zv = GV(z_t, (i_)); const double *seg_ = Rt + (size_t)rowstart((i_), d) * 64; \
    const int ui_ = (i_) - J0, m_ = d - 1 - (i_); \
                                   _Pragma("unroll") for (int u = 0; u < TW; ++u) { int k = u - ui_; k = k < 0 ? 0 : k; k = k > m_ ? m_ : k; rv[u] = LDNT(seg_, k); } }
#define MCX_TRMV_FMD(rv, zv, i_) { const int ui_ = (i_) - J0; _Pragma("unroll") for (int u = 0; u < TW; ++u) { double nv = dfma(rv[u], zv, P[u]); P[u] = (u >= ui_) ? nv : P[u]; } }
                    if (PIPE) {
                        MCX_TRMV_LDD(da, za, J0)
                        for (int i = J0; i < J0 + nw; i += 2) {
                            if (i + 1 < J0 + nw) MCX_TRMV_LDD(db, zb, i + 1)
                            MCX_TRMV_FMD(da, za, i)
                            if (i + 2 < J0 + nw) MCX_TRMV_LDD(da, za, i + 2)
                            if (i + 1 < J0 + nw) MCX_TRMV_FMD(db, zb, i + 1)
                        }
                    } else {
                        for (int i = J0; i < J0 + nw; ++i) { MCX_TRMV_LDD(da, za, i) MCX_TRMV_FMD(da, za, i) }
                    }
#undef MCX_TRMV_LDD
#undef MCX_TRMV_FMD
                }
                double th[TW];                           // the state's loads before the candidate's stores (see copy_vec)
#pragma unroll
                for (int u = 0; u < TW; ++u) th[u] = GV(theta_t, J0 + (u < nw ? u : nw - 1));
#pragma unroll
                for (int u = 0; u < TW; ++u) if (u < nw) GV(P_t, J0 + u) = th[u] + P[u];   // newpar = oldpar + R'z
            }
        }
    }
    // The proposal that follows a successful Cholesky downdate accumulates from the diagonal up (mcxo_trmv_ut_desc): this
    // standalone form serves the cases where ram_update could not fuse it (first iteration of a launch, host callbacks).
    if (__any(dsc)) {
        for (int J0 = 0; J0 < d; J0 += TW) {
            const int nw = (d - J0) < TW ? (d - J0) : TW;
            double P[TW];
#pragma unroll
            for (int u = 0; u < TW; ++u) P[u] = 0.0;
            if (dsc) {
                for (int i = J0 + nw - 1; i >= J0; --i) {                    // diagonal block, rows descending
                    const double zi = GV(z_t, i);
                    const double *seg = Rt + (size_t)rowstart(i, d) * 64;
                    const int ui = i - J0, m = d - 1 - i;
                    double r[TW];
#pragma unroll
                    for (int u = 0; u < TW; ++u) { int k = u - ui; k = k < 0 ? 0 : k; k = k > m ? m : k; r[u] = LDNT(seg, k); }
#pragma unroll
                    for (int u = 0; u < TW; ++u) {
                        const double nv = (u == ui) ? r[u] * zi : dfma(r[u], zi, P[u]);
                        P[u] = (u >= ui) ? nv : P[u];
                    }
                }
#pragma unroll 2
                for (int i = J0 - 1; i >= 0; --i) {                          // rows above, descending
                    const double zi = GV(z_t, i);
                    const double *seg = Rt + (size_t)(rowstart(i, d) + J0 - i) * 64;
                    double r[TW];
#pragma unroll
                    for (int u = 0; u < TW; ++u) r[u] = LDNT(seg, u < nw ? u : nw - 1);
#pragma unroll
                    for (int u = 0; u < TW; ++u) P[u] = dfma(r[u], zi, P[u]);
                }
                double th[TW];
#pragma unroll
                for (int u = 0; u < TW; ++u) th[u] = GV(theta_t, J0 + (u < nw ? u : nw - 1));
#pragma unroll
                for (int u = 0; u < TW; ++u) if (u < nw) GV(P_t, J0 + u) = th[u] + P[u];
            }
        }
    }
}

// ---------------------------------------------------------------- full-matrix products for the SVD paths
// y = M x (dgemv 'N', matutils.F90:161): y = 0, then column by column y_i += x_j M(i,j) -- each y_i is an fma chain
// ascending in j.  Row panels of PW accumulators in registers; out_t = (add_t ? add_t : 0) + y.
// PIPE (the per-chain SCAM kernel, which does nothing but stream its rotation): four columns' loads in flight -- left to
// itself the compiler sinks every load next to its fma and keeps ~4 outstanding.  The step kernels (SVD proposal factor)
// keep the plain form: their registers are spoken for.
template <bool PIPE = false, int NBO = 0>       // NBO: rows in flight, when not the default of PIPE
MCX_DEV void gemvN_panels(const double *Mt, const double *x_t, double *out_t, const double *add_t, int lane, int d, bool act,
                          int p0 = 0, int pstep = 1)                 // panels p0, p0 + pstep, ...: a workgroup's waves share the rows
{
#ifndef MCX_GEMV_NB
#define MCX_GEMV_NB 4
#endif
    constexpr int NB = NBO ? NBO : (PIPE ? MCX_GEMV_NB : 1);
    for (int I0 = p0 * PW; I0 < d; I0 += pstep * PW) {
        const int nr = (d - I0) < PW ? (d - I0) : PW;
        double y[PW];
#pragma unroll
        for (int u = 0; u < PW; ++u) y[u] = 0.0;
        if (act) {
            double rr[NB][PW], xx[NB];                             // the matrix is streamed (non-temporal), x stays cached
#define MCX_GEMV_LD(s_, j_) { xx[s_] = GV(x_t, (j_)); const double *seg_ = Mt + ((size_t)(j_) * d + I0) * 64; \
                              _Pragma("unroll") for (int u = 0; u < PW; ++u) rr[s_][u] = LDNT(seg_, u < nr ? u : nr - 1); }
#pragma unroll
            for (int s = 0; s < NB - 1; ++s) if (s < d) MCX_GEMV_LD(s, s)
#pragma unroll (NB == 1 ? 4 : 1)
            for (int j = 0; j < d; j += NB) {
#pragma unroll
                for (int s = 0; s < NB; ++s) {
                    if (j + s + NB - 1 < d) MCX_GEMV_LD((s + NB - 1) % NB, j + s + NB - 1)
                    if (j + s < d) {
#pragma unroll
                        for (int u = 0; u < PW; ++u) y[u] = dfma(xx[s], rr[s][u], y[u]);
                    }
                }
            }
#undef MCX_GEMV_LD
#pragma unroll
            for (int u = 0; u < PW; ++u) if (u < nr) GV(out_t, I0 + u) = add_t ? (GV(add_t, I0 + u) + y[u]) : y[u];
        }
    }
}
// y = M'x (dgemv 'T'): y_k = sum_i M(i,k) x_i, i ascending, one fma chain per column.
template <bool PIPE = false, int NBO = 0>
MCX_DEV void gemvT_panels(const double *Mt, const double *x_t, double *out_t, int lane, int d, int p0 = 0, int pstep = 1)
{
#ifndef MCX_GEMV_NB
#define MCX_GEMV_NB 4
#endif
    constexpr int NB = NBO ? NBO : (PIPE ? MCX_GEMV_NB : 1);
    for (int K0 = p0 * PW; K0 < d; K0 += pstep * PW) {
        const int nc = (d - K0) < PW ? (d - K0) : PW;
        double t[PW];
#pragma unroll
        for (int u = 0; u < PW; ++u) t[u] = 0.0;
        double rr[NB][PW], xx[NB];
#define MCX_GEMV_LD(s_, i_) { xx[s_] = GV(x_t, (i_)); _Pragma("unroll") for (int u = 0; u < PW; ++u) rr[s_][u] = LDNT(Mt, (size_t)(K0 + (u < nc ? u : nc - 1)) * d + (i_)); }
#pragma unroll
        for (int s = 0; s < NB - 1; ++s) if (s < d) MCX_GEMV_LD(s, s)
#pragma unroll (NB == 1 ? 4 : 1)
        for (int i = 0; i < d; i += NB) {
#pragma unroll
            for (int s = 0; s < NB; ++s) {
                if (i + s + NB - 1 < d) MCX_GEMV_LD((s + NB - 1) % NB, i + s + NB - 1)
                if (i + s < d) {
#pragma unroll
                    for (int u = 0; u < PW; ++u) t[u] = dfma(rr[s][u], xx[s], t[u]);
                }
            }
        }
#undef MCX_GEMV_LD
#pragma unroll
        for (int u = 0; u < PW; ++u) if (u < nc) GV(out_t, K0 + u) = t[u];
    }
}

// The routine's dot products (oracle/mcx_svd.h): eight partial fma chains over the rows k = j, j + 8, ... and the pairwise tree
MCX_DEV double svd_tree8(const double (&p)[8]) { return ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7])); }

// The pinned dgesvd('A','N') of a symmetric PSD matrix: one-sided Jacobi, row-cyclic, operation for operation the
// routine of oracle/mcx_svd.h (see there).  Gt: in the matrix (column-major d*d per chain), destroyed; Vt: out
// the singular vectors; sv_t: out singular values, descending.  Lanes converge independently; a converged lane
// keeps re-deriving "no rotation" from unchanged data, which is the same as having left the loop.
MCX_DEV void symsvd_dev(double *Gt, double *Vt, double *sv_t, int lane, int d, bool act)
{
    for (int j = 0; j < d; ++j) for (int i = 0; i < d; ++i) if (act) GV(Vt, (size_t)j * d + i) = (i == j) ? 1.0 : 0.0;
    // One pass over k per pair: the rotation of (g_p, g_q) and (v_p, v_q) and, on the fly, the three dot products of
    // the NEXT pair (p, q+1), which see g_p as this rotation leaves it.  Every chain of operations is the one of
    // oracle/mcx_svd.h (same operands, same order); only the loops are merged, so that a pair costs one latency-bound
    // sweep over the columns instead of three.  A pair nobody in the wave rotates leaves g_p alone: alpha carries over
    // (the same fma chain over the same data), beta and gamma of the next pair take one read of g_p and g_{q+1}.
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < d - 1; ++p) {
            double *gp = Gt + (size_t)p * d * 64, *vp = Vt + (size_t)p * d * 64;
            double alpha, beta, gamma;
            {
                const double *gq = gp + (size_t)d * 64;
                double pa[8], pb[8], pg[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { pa[u] = 0.0; pb[u] = 0.0; pg[u] = 0.0; }
                for (int k0 = 0; k0 < d; k0 += 8) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (k0 + u < d) {
                            double a = GV(gp, k0 + u), b = GV(gq, k0 + u);
                            pa[u] = dfma(a, a, pa[u]); pb[u] = dfma(b, b, pb[u]); pg[u] = dfma(a, b, pg[u]);
                        }
                    }
                }
                alpha = svd_tree8(pa); beta = svd_tree8(pb); gamma = svd_tree8(pg);
            }
            for (int q = p + 1; q < d; ++q) {
                double *gq = Gt + (size_t)q * d * 64, *vq = Vt + (size_t)q * d * 64;
                const bool more = q + 1 < d;
                const double *gn = more ? gq + (size_t)d * 64 : gq;          // column q+1 (unused when !more)
                const bool rot = act && (gamma != 0.0) && !(fabs(gamma) <= 1e-15 * sqrt(alpha * beta));
                double na = 0.0, nb = 0.0, ng = 0.0;
                double pa[8], pb[8], pg[8];                      // partial chains by row index mod 8 (SB = 8 rows per block below)
#pragma unroll
                for (int u = 0; u < 8; ++u) { pa[u] = 0.0; pb[u] = 0.0; pg[u] = 0.0; }
                if (__any(rot)) {
                    double c = 1.0, sn = 0.0;
                    if (rot) {
                        rotated = true;
                        double zeta = (beta - alpha) / (2.0 * gamma);
                        double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                        c = 1.0 / sqrt(1.0 + t * t); sn = c * t;
                    }
                    // blocks of SB rows, the next block's fifteen loads issued before this block's stores: the
                    // columns alias as far as the compiler can tell, so without this every row would wait for its own loads
                    constexpr int SB = 8;
                    double A[SB], B[SB], Ee[SB], VA[SB], VB[SB];
#pragma unroll
                    for (int u = 0; u < SB; ++u) {
                        const int k = u < d ? u : d - 1;
                        A[u] = GV(gp, k); B[u] = GV(gq, k); Ee[u] = GV(gn, k); VA[u] = GV(vp, k); VB[u] = GV(vq, k);
                    }
                    for (int k0 = 0; k0 < d; k0 += SB) {
                        double A2[SB], B2[SB], E2[SB], VA2[SB], VB2[SB];
#pragma unroll
                        for (int u = 0; u < SB; ++u) {
                            int k = k0 + SB + u; k = k < d ? k : d - 1;
                            A2[u] = GV(gp, k); B2[u] = GV(gq, k); E2[u] = GV(gn, k); VA2[u] = GV(vp, k); VB2[u] = GV(vq, k);
                        }
#pragma unroll
                        for (int u = 0; u < SB; ++u) {
                            const int k = k0 + u;
                            if (k < d) {
                                const double a = A[u], b = B[u], e = Ee[u], va = VA[u], vb = VB[u];
                                const double ra = c * a - sn * b, rb = sn * a + c * b;
                                const double aa = rot ? ra : a;
                                if (rot) { GV(gp, k) = ra; GV(gq, k) = rb; GV(vp, k) = c * va - sn * vb; GV(vq, k) = sn * va + c * vb; }
                                pa[u] = dfma(aa, aa, pa[u]); pb[u] = dfma(e, e, pb[u]); pg[u] = dfma(aa, e, pg[u]);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < SB; ++u) { A[u] = A2[u]; B[u] = B2[u]; Ee[u] = E2[u]; VA[u] = VA2[u]; VB[u] = VB2[u]; }
                    }
                    na = svd_tree8(pa); nb = svd_tree8(pb); ng = svd_tree8(pg);
                } else if (more) {
                    na = alpha;
                    for (int k0 = 0; k0 < d; k0 += 8) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            if (k0 + u < d) {
                                const double a = GV(gp, k0 + u), e = GV(gn, k0 + u);
                                pb[u] = dfma(e, e, pb[u]); pg[u] = dfma(a, e, pg[u]);
                            }
                        }
                    }
                    nb = svd_tree8(pb); ng = svd_tree8(pg);
                }
                alpha = na; beta = nb; gamma = ng;
            }
        }
        if (!__any(rotated)) break;
    }
    if (act) {
        for (int j = 0; j < d; ++j) {
            const double *gj = Gt + (size_t)j * d * 64;
            double pa[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) pa[u] = 0.0;
            for (int k0 = 0; k0 < d; k0 += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) if (k0 + u < d) { double g = GV(gj, k0 + u); pa[u] = dfma(g, g, pa[u]); }
            }
            GV(sv_t, j) = sqrt(svd_tree8(pa));
        }
        for (int i = 0; i < d - 1; ++i) {                     // descending order, first maximum wins
            int m = i; double sm = GV(sv_t, i);
            for (int j = i + 1; j < d; ++j) { double sj = GV(sv_t, j); if (sj > sm) { m = j; sm = sj; } }
            if (m != i) {
                double ts = GV(sv_t, i); GV(sv_t, i) = GV(sv_t, m); GV(sv_t, m) = ts;
                for (int k = 0; k < d; ++k) {
                    double tv = GV(Vt, (size_t)i * d + k); GV(Vt, (size_t)i * d + k) = GV(Vt, (size_t)m * d + k); GV(Vt,
                        (size_t)m * d + k) = tv;
                }
            }
        }
    }
}

// Same product with ONE factor shared by every chain (pooled mode): the factor is wave-uniform, so its
// elements come through the scalar cache (s_load) and the only vector traffic is the chain's own z and P.
MCX_DEV void trmv_shared(const double *__restrict__ Rs, const double *z_t, double *P_t, const double *theta_t, int lane, int d)
{
    for (int J0 = 0; J0 < d; J0 += PW) {
        const int nw = (d - J0) < PW ? (d - J0) : PW;
        double P[PW];
#pragma unroll
        for (int u = 0; u < PW; ++u) P[u] = 0.0;
#pragma unroll 2
        for (int i = 0; i < J0; ++i) {
            const double zi = GV(z_t, i);
            const double *__restrict__ seg = Rs + (size_t)(rowstart(i, d) + J0 - i);
#pragma unroll
            for (int u = 0; u < PW; ++u) P[u] = dfma(seg[u < nw ? u : nw - 1], zi, P[u]);
        }
        for (int i = J0; i < J0 + nw; ++i) {
            const double zi = GV(z_t, i);
            const double *__restrict__ seg = Rs + (size_t)rowstart(i, d);
            const int ui = i - J0, m = d - 1 - i;
#pragma unroll
            for (int u = 0; u < PW; ++u) {
                int k = u - ui; k = k < 0 ? 0 : k; k = k > m ? m : k;
                double nv = dfma(seg[k], zi, P[u]);
                P[u] = (u >= ui) ? nv : P[u];
            }
        }
        double th[PW];
#pragma unroll
        for (int u = 0; u < PW; ++u) th[u] = GV(theta_t, J0 + (u < nw ? u : nw - 1));
#pragma unroll
        for (int u = 0; u < PW; ++u) if (u < nw) GV(P_t, J0 + u) = th[u] + P[u];
    }
}

// matmulx(R, z) with ONE full factor shared by every chain (pooled mode with condmax > 0): M[j*d + i] = R(i,j) (column-major,
// padded by PWS doubles), y_i an fma chain ascending in j like gemvN_panels; the matrix comes through the scalar cache.
MCX_DEV void gemvN_shared(const double *__restrict__ M, const double *z_t, double *out_t, const double *theta_t, int lane, int d)
{
    for (int I0 = 0; I0 < d; I0 += PW) {
        const int nr = (d - I0) < PW ? (d - I0) : PW;
        double y[PW];
#pragma unroll
        for (int u = 0; u < PW; ++u) y[u] = 0.0;
#pragma unroll 2
        for (int j = 0; j < d; ++j) {
            const double zj = GV(z_t, j);
            const double *__restrict__ col = M + (size_t)j * d + I0;
#pragma unroll
            for (int u = 0; u < PW; ++u) y[u] = dfma(zj, col[u < nr ? u : nr - 1], y[u]);
        }
        double th[PW];
#pragma unroll
        for (int u = 0; u < PW; ++u) th[u] = GV(theta_t, I0 + (u < nr ? u : nr - 1));
#pragma unroll
        for (int u = 0; u < PW; ++u) if (u < nr) GV(out_t, I0 + u) = th[u] + y[u];
    }
}

} // namespace mcx
