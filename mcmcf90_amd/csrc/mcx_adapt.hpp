// mcx_adapt.hpp -- the first point (init_kernel) and MCMC_adapt at a tick (MCMC_adapt.F90:12-230): schedule, covmat in 10 x 10 blocks,
// MCMC_calculate_R (one of the family headers mcx_kernels.hpp includes, in this order: mcx_common, mcx_products, mcx_step, mcx_scam,
// mcx_pooled, mcx_phase, mcx_adapt, mcx_svd, mcx_moments)
#pragma once
#include "mcx_phase.hpp"

namespace mcx {

// ---------------------------------------------------------------- first point (MCMC_run.F90:33-39)
// dst[(tile*K + e)*64 + lane] = src[e]: every chain starts from the same K-vector (par0, R(cmat0), ...)
__global__ __launch_bounds__(64) void bcast_kernel(double *dst, const double *__restrict__ src, size_t K)
{
    double *o = dst + (size_t)blockIdx.x * K * 64;
    for (size_t e = blockIdx.y; e < K; e += gridDim.y) o[e * 64 + threadIdx.x] = src[e];
}

__global__ __launch_bounds__(64) void init_kernel(EngineDev E)
{
    const int lane = threadIdx.x, tile = blockIdx.x, d = E.d;
    double *theta_t = E.theta + (size_t)tile * d * 64;
    double pri1, ss1;
    const int ny = E.ny;
    if (E.tgt.kind == TGT_HOST) {
        const double *hev = E.hev + (size_t)tile * (NHE - 1 + ny) * 64;
        pri1 = GV(hev, HE_PRI); ss1 = GV(hev, HE_SS);
        for (int j = 0; j < (ny > 1 ? ny : 0); ++j) TIDX(E.ssv, tile, ny, j, lane) = GV(hev, HE_SS + j);
    }
    else { pri1 = target_prior(E.tgt, d, lane, theta_t); ss1 = target_ss<false>(E.tgt, d, lane, theta_t, E.tgt.mu, E.tgt.lamT); }
    TIDX(E.scal, tile, NSCAL, S_SS1, lane) = ss1; TIDX(E.scal, tile, NSCAL, S_PRI1, lane) = pri1;
    // row 1 of the chain: iteration 1 counts as accepted
    const int slot = 1 % E.wcap;
    if (E.hist) {
        double *h = E.hist + ((size_t)tile * E.wcap + slot) * (size_t)E.hs * 64;
        for (int k = 0; k < d; ++k) GV(h, k) = GV(theta_t, k);
        GV(h, d) = ss1;
        for (int j = 1; j < ny; ++j) GV(h, d + j) = TIDX(E.ssv, tile, ny, j, lane);
        if (lane == 0) E.wacc[(size_t)tile * E.wcap + slot] = ~0ull;
        if (E.record_s2) {
            if (ny > 1) { for (int j = 0; j < ny; ++j) E.s2hist[(((size_t)tile * E.wcap + slot) * ny + j) * 64 + lane] = TIDX(E.s2v, tile,
                ny, j, lane); }
            else E.s2hist[((size_t)tile * E.wcap + slot) * 64 + lane] = TIDX(E.scal, tile, NSCAL, S_SIGMA2, lane);
        }
    }
    if (E.accmask && lane == 0) E.accmask[tile] = ~0ull;
    for (int k = 0; k < d; ++k) TIDX(E.basetheta, tile, d, k, lane) = GV(theta_t, k);
}

// ---------------------------------------------------------------- MCMC_adapt (MCMC_adapt.F90:12-174) at a tick
// mode bits chosen by the host from (simuind, namelist): see mcmcx_run.
enum { AD_BURN = 1, AD_AM = 2, AD_FIRST = 4 };

// MCMC_calculate_R, Cholesky branch (MCMC_adapt.F90:211-215): R = dpotf2('U', cmat) * 2.4/sqrt(d).
// dpotf2 computes R(j,k) = (A(j,k) - sum_{i<j} R(i,j) R(i,k)) / R(j,j) with every sum an fma chain ascending in i
// from 0.  Same chains here, but formed for an 8 x 8 block of (j,k) at a time: the 64 accumulators stay in registers
// while the finished rows i < J0 stream by once per block (16 loads per 64 fma instead of 1 per fma), then the rows
// of the block row itself are folded in -- from registers on the diagonal block, whose finished rows and 1/R(j,j) are
// parked in LDS for the blocks to its right.  At: cmat (read), Tt: the factor (written, and read back as rows i < J0),
// Rt: scaled copy on success.  X: 36 LDS vectors.  Returns LAPACK's info (0, or j+1 at the first non-positive pivot).
constexpr int BT = 8;
#define MCX_DLI(a, b) ((a) * (15 - (a)) / 2 + ((b) - (a) - 1))   // strictly upper part of the 8 x 8 diagonal block, by rows: 28 entries
MCX_DEV int calculate_R(const double *At, double *Tt, double *Rt, int lane, int d, int P, bool act, double *X)
{
    int info = 0;
    for (int J0 = 0; J0 < d; J0 += BT) {
        const int nr = (d - J0) < BT ? (d - J0) : BT;
        for (int K0 = J0; K0 < d; K0 += BT) {
            const int nc = (d - K0) < BT ? (d - K0) : BT;
            const bool diag = (K0 == J0);
            double T[BT][BT];
#pragma unroll
            for (int a = 0; a < BT; ++a)
#pragma unroll
                for (int b = 0; b < BT; ++b) T[a][b] = 0.0;
#pragma unroll 2
            for (int i = 0; i < J0; ++i) {
                const double *rowi = Tt + (size_t)rowstart(i, d) * 64;             // element (i,k) at rowi[k - i]
                double rj[BT], rk[BT];
#pragma unroll
                for (int a = 0; a < BT; ++a) rj[a] = GV(rowi, J0 - i + (a < nr ? a : nr - 1));
#pragma unroll
                for (int b = 0; b < BT; ++b) rk[b] = GV(rowi, K0 - i + (b < nc ? b : nc - 1));
#pragma unroll
                for (int a = 0; a < BT; ++a)
#pragma unroll
                    for (int b = 0; b < BT; ++b) T[a][b] = dfma(rj[a], rk[b], T[a][b]);
            }
#pragma unroll
            for (int a = 0; a < BT; ++a) {
                if (a < nr) {
                    const int j = J0 + a;
                    const double *arow = At + (size_t)rowstart(j, d) * 64;
                    double *trow = Tt + (size_t)rowstart(j, d) * 64;
                    double av[BT];
#pragma unroll
                    for (int b = 0; b < BT; ++b) { int k = K0 + (b < nc ? b : nc - 1); av[b] = GV(arow, (k >= j ? k : j) - j); }
                    if (diag) {
#pragma unroll
                        for (int a2 = 0; a2 < a; ++a2)
#pragma unroll
                            for (int b = a; b < BT; ++b) T[a][b] = dfma(T[a2][a], T[a2][b], T[a][b]);
                        const double ajj = av[a] - T[a][a];
                        if (act && info == 0 && !(ajj > 0.0)) info = j + 1;
                        const double rjj = sqrt(ajj), rinv = 1.0 / rjj;
                        T[a][a] = rjj;
#pragma unroll
                        for (int b = a + 1; b < BT; ++b) T[a][b] = (av[b] - T[a][b]) * rinv;
#pragma unroll
                        for (int b = a; b < BT; ++b) {
                            if (b < nc) { GV(trow, K0 + b - j) = T[a][b]; if (b > a) X[MCX_DLI(a, b) * 64 + lane] = T[a][b]; }
                        }
                        X[(28 + a) * 64 + lane] = rinv;
                    } else {
#pragma unroll
                        for (int a2 = 0; a2 < a; ++a2) {
                            const double dl = X[MCX_DLI(a2, a) * 64 + lane];
#pragma unroll
                            for (int b = 0; b < BT; ++b) T[a][b] = dfma(dl, T[a2][b], T[a][b]);
                        }
                        const double rinv = X[(28 + a) * 64 + lane];
#pragma unroll
                        for (int b = 0; b < BT; ++b) {
                            T[a][b] = (av[b] - T[a][b]) * rinv;
                            if (b < nc) GV(trow, K0 + b - j) = T[a][b];
                        }
                    }
                }
            }
        }
    }
    if (act && info == 0) {
        double sq = sqrt((double)d);
        map_vec(Rt, Tt, lane, P, [&](double v) { return v * 2.4 / sq; });
    }
    return info;
}
#undef MCX_DLI

// dpotri('U') on a packed upper factor, in place: dtrti2('U','N') then dlauu2('U') (MCMC_adapt.F90:217-224).
// On exit A holds the upper triangle of inv(R'R).  X (LDS) carries one column above the diagonal.
// X += temp * column(k) over the rows r < n, four rows at a time: the column's loads and X's go out together, then the four
// independent fmas (each element's own chain is unchanged) -- the plain loop is a load-fma-store round trip per row, because
// the compiler must assume the vector and the matrix overlap
MCX_DEV void potri_axpy_col(double *X, const double *At, int lane, int d, int n, int k, double temp)
{
    // eight rows per trip, the last trip's spare slots re-read row n - 1 and are dropped: a trip is one cache round trip, and dpotri is
    // ~1700 of them in a row at npar 20 (round 4; four rows per trip plus an element-by-element tail before)
    constexpr int NB = 8;
    for (int r = 0; r < n; r += NB) {
        double a[NB], x[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) { const int ru = (r + u < n) ? r + u : n - 1; a[u] = GV(At, pidx(ru, k, d)); x[u] = XL(ru); }
#pragma unroll
        for (int u = 0; u < NB; ++u) if (r + u < n) XL(r + u) = dfma(temp, a[u], x[u]);
    }
}
MCX_DEV int potri_packed(double *At, int lane, int d, bool act, double *X)
{
    int info = 0;
    for (int j = 0; j < d; ++j) if (act && info == 0 && GV(At, pidx(j, j, d)) == 0.0) info = j + 1;
    const bool go = act && info == 0;
    if (__any(go)) {
        if (go) {
            for (int j = 0; j < d; ++j) {                    // dtrti2
                double ajj = 1.0 / GV(At, pidx(j, j, d));
                GV(At, pidx(j, j, d)) = ajj;
                ajj = -ajj;
                // (eight loads in flight: element by element every one is a cache round trip)
                for (int i0 = 0; i0 < j; i0 += 8) {
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = GV(At, pidx((i0 + u < j) ? i0 + u : j - 1, j, d));
#pragma unroll
                    for (int u = 0; u < 8; ++u) if (i0 + u < j) XL(i0 + u) = v[u];
                }
                for (int j0 = 0; j0 < j; j0 += 8) {          // dtrmv('U','N','N') with the inverted leading block
                    double dg[8];                            // (its diagonal: eight loads in flight)
#pragma unroll
                    for (int u = 0; u < 8; ++u) dg[u] = GV(At, pidx((j0 + u < j) ? j0 + u : j - 1, (j0 + u < j) ? j0 + u : j - 1, d));
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int jj = j0 + u;
                        if (jj < j) {
                            double temp = XL(jj);
                            if (temp != 0.0) {
                                potri_axpy_col(X, At, lane, d, jj, jj, temp);
                                XL(jj) = temp * dg[u];
                            }
                        }
                    }
                }
                for (int i = 0; i < j; ++i) GV(At, pidx(i, j, d)) = ajj * XL(i);
            }
            for (int i = 0; i < d; ++i) {                    // dlauu2
                double *rowi = At + (size_t)rowstart(i, d) * 64;
                double aii = GV(rowi, 0);
                if (i < d - 1) {
                    double dot = 0.0;
                    for (int k0 = 0; k0 < d - i; k0 += 8) {
                        double v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = GV(rowi, (k0 + u < d - i) ? k0 + u : d - i - 1);
#pragma unroll
                        for (int u = 0; u < 8; ++u) if (k0 + u < d - i) dot = dfma(v[u], v[u], dot);
                    }
                    GV(rowi, 0) = dot;
                    for (int r0 = 0; r0 < i; r0 += 8) {
                        double v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = GV(At, pidx((r0 + u < i) ? r0 + u : i - 1, i, d));
#pragma unroll
                        for (int u = 0; u < 8; ++u) if (r0 + u < i) XL(r0 + u) = aii * v[u];
                    }
                    for (int k0 = i + 1; k0 < d; k0 += 8) {
                        double tv[8];                        // (row i's elements: eight loads in flight)
#pragma unroll
                        for (int u = 0; u < 8; ++u) tv[u] = GV(rowi, ((k0 + u < d) ? k0 + u : d - 1) - i);
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int k = k0 + u;
                            if (k < d && tv[u] != 0.0) potri_axpy_col(X, At, lane, d, i, k, tv[u]);
                        }
                    }
                    for (int r = 0; r < i; ++r) GV(At, pidx(r, i, d)) = XL(r);
                } else {
                    for (int r = 0; r <= i; ++r) GV(At, pidx(r, i, d)) = aii * GV(At, pidx(r, i, d));
                }
            }
        }
    }
    return info;
}

// covmat (matutils.F90:232-341) over the nr rows listed in `rows` (ring slot | weight << 32; slot 0xffffffff =
// the window's base row in basetheta) for one chain per lane.  update && wsum > 0: weighted Welford, one row
// at a time (:283-310); otherwise the two-pass batch branch (:311-338), which overwrites cmat, mean and wsum.
MCX_DEV void covmat_rows(const EngineDev &E, int tile, int lane, const uint64_t *rows, int nr, bool act, bool update,
                         double *Ct, double *mean_t, const double *base_t, double *m2_t, double &wsum, double *X)
{
    const int d = E.d, P = E.P;
    int nrmax = act ? nr : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int other = __shfl_xor(nrmax, o); nrmax = other > nrmax ? other : nrmax; }
    const double *hist_t = E.hist + (size_t)tile * E.wcap * (size_t)E.hs * 64;
    const bool upd = act && update && (wsum > 0.0);
    const bool bat = act && !upd;
    if (__any(upd)) {
        for (int r = 0; r < nrmax; ++r) {
            const bool on = upd && r < nr;
            uint64_t e = on ? GV(rows, r) : 0ull;
            uint32_t slot = (uint32_t)e; double w3 = (double)(uint32_t)(e >> 32);
            if (on) {
                const bool isbase = (slot == 0xffffffffu);
                const size_t so = isbase ? 0 : (size_t)slot * (size_t)E.hs * 64;
                for (int k = 0; k < d; ++k) {
                    double xv = isbase ? GV(base_t, k) : hist_t[so + (size_t)k * 64 + lane];
                    XL(k) = xv - GV(mean_t, k);
                }
                double f1 = w3 / (wsum + w3 - 1.0);
                double f2 = wsum / (wsum + w3);
                for (int a = 0; a < d; ++a) {             // row a of the upper triangle: elements (a, b >= a)
                    double da = XL(a);
                    double *rowa = Ct + (size_t)rowstart(a, d) * 64;
                    const int n = d - a;
                    for (int k0 = 0; k0 < n; k0 += CH) {  // CH elements per batch: loads first, then the updates
                        double cab[CH];
#pragma unroll
                        for (int u = 0; u < CH; ++u) cab[u] = GV(rowa, (k0 + u < n) ? k0 + u : n - 1);
#pragma unroll
                        for (int u = 0; u < CH; ++u) {
                            if (k0 + u < n) {
                                double o = da * XL(a + k0 + u);
                                GV(rowa, k0 + u) = cab[u] + f1 * (f2 * o - cab[u]);
                            }
                        }
                    }
                }
                double f3 = w3 / (wsum + w3);
                for (int k = 0; k < d; ++k) GV(mean_t, k) = GV(mean_t, k) + f3 * XL(k);
                wsum = w3 + wsum;
            }
        }
    }
    if (__any(bat)) {
        double wsum2 = 0.0;
        if (bat) {
            for (int r = 0; r < nr; ++r) wsum2 = wsum2 + (double)(uint32_t)(GV(rows, r) >> 32);
            for (int k = 0; k < d; ++k) GV(m2_t, k) = 0.0;
        }
        for (int r = 0; r < nrmax; ++r) {
            const bool on = bat && r < nr;
            uint64_t e = on ? GV(rows, r) : 0ull;
            uint32_t slot = (uint32_t)e; double w = (double)(uint32_t)(e >> 32);
            if (on) {
                const bool isbase = (slot == 0xffffffffu);
                const size_t so = isbase ? 0 : (size_t)slot * (size_t)E.hs * 64;
                for (int k = 0; k < d; ++k) {
                    double xv = isbase ? GV(base_t, k) : hist_t[so + (size_t)k * 64 + lane];
                    GV(m2_t, k) = GV(m2_t, k) + xv * w;
                }
            }
        }
        if (bat) {
            for (int k = 0; k < d; ++k) GV(m2_t, k) = GV(m2_t, k) / wsum2;          // xmean2
            for (int e = 0; e < P; ++e) GV(Ct, e) = 0.0;
        }
        for (int r = 0; r < nrmax; ++r) {
            const bool on = bat && r < nr;
            uint64_t e = on ? GV(rows, r) : 0ull;
            uint32_t slot = (uint32_t)e; double w = (double)(uint32_t)(e >> 32);
            if (on) {
                const bool isbase = (slot == 0xffffffffu);
                const size_t so = isbase ? 0 : (size_t)slot * (size_t)E.hs * 64;
                for (int k = 0; k < d; ++k) {
                    double xv = isbase ? GV(base_t, k) : hist_t[so + (size_t)k * 64 + lane];
                    XL(k) = xv - GV(m2_t, k);
                }
                // reference: cmat(i,j), j <= i = sum_r (x_ri - m_i) * ((x_rj - m_j) * w_r); kept at packed (j,i)
                for (int j = 0; j < d; ++j) {
                    double xb = XL(j) * w;
                    double *rowj = Ct + (size_t)rowstart(j, d) * 64;
                    const int n = d - j;
                    for (int k0 = 0; k0 < n; k0 += CH) {
                        double cji[CH];
#pragma unroll
                        for (int u = 0; u < CH; ++u) cji[u] = GV(rowj, (k0 + u < n) ? k0 + u : n - 1);
#pragma unroll
                        for (int u = 0; u < CH; ++u) if (k0 + u < n) GV(rowj, k0 + u) = cji[u] + XL(j + k0 + u) * xb;
                    }
                }
            }
        }
        if (bat) {
            map_vec(Ct, Ct, lane, P, [&](double v) { return v / (wsum2 - 1.0); });
            copy_vec(mean_t, m2_t, nullptr, lane, d);
            wsum = wsum2;
        }
    }
}


// One MCMC_adapt tick is a handful of launches.  adapt_pre_kernel (one wave per tile) runs the schedule's branch up to the covariance
// update: burn-in scaling, the greedy / first-tick restarts, the list of window rows.  adapt_cov_diag_kernel / adapt_cov_off_kernel run the
// steady-state Welford update with ONE 10 x 10 block of chaincmat per wave (covmat_window_td below): the blocks of a tile are separate
// workgroups that walk the same window of the history ring at about the same time, laid out over the grid so that they land on the same
// XCD (workgroups go round-robin over the 8 XCDs) -- the window is fetched from HBM once and served to the other blocks by that XCD's L2.
// adapt_covb_* do the same for covmat's two-pass batch branch.  adapt_post_kernel (one wave per tile) finishes: the one-off batch branches
// lane by lane (covmat_rows), the window restart and MCMC_calculate_R -- or hands the factorisation to tile_factor_kernel (mcx_group.hpp)
// or to the blocked SVD (mcx_svd.hpp).  Every element of chaincmat / chainmean sees the operations of the reference's covmat.
enum { ADF_DOCALC = 1, ADF_GREEDY = 2, ADF_STEADY = 4,
       // the lane takes covmat's two-pass batch branch over the row list (first AM adaptation with initcmatn = 0, AP window, greedy restart
       // with initcmatn = 0); I_BSTART = the first iteration whose ballot belongs to the list
       ADF_BATCH = 8,
       // ... and the list's first row is the one accepted AT I_BSTART (greedy: rows 1..it) instead of a row from before it
       ADF_BNOINIT = 16 };

__global__ __launch_bounds__(64) void adapt_pre_kernel(EngineDev E, int it, int mode)
{
    const int lane = threadIdx.x, tile = blockIdx.x, d = E.d, P = E.P;
    double *Rt = E.R + (size_t)tile * P * 64;
    double *Ct = E.cmat + (size_t)tile * P * 64;
    double *mean_t = E.mean + (size_t)tile * d * 64;
    uint64_t *rows = E.rowlist + (size_t)tile * (E.wcap + 1) * 64;
    uint32_t stayed = TIDX(E.ictr, tile, NICTR, I_STAYED, lane);
    uint32_t curcount = TIDX(E.ictr, tile, NICTR, I_CURCOUNT, lane);
    uint32_t lastfreq = TIDX(E.ictr, tile, NICTR, I_LASTFREQ, lane);
    uint32_t basecnt = TIDX(E.ictr, tile, NICTR, I_BASECNT, lane);
    uint32_t winstart = TIDX(E.ictr, tile, NICTR, I_WINSTART, lane);
    double wsum = TIDX(E.scal, tile, NSCAL, S_WSUM, lane);
    uint32_t flags = 0, bstart = 0;
    int nr = 0;

    if (mode & AD_BURN) {                                             // MCMC_adapt.F90:60-102
        double staypc = (double)stayed / (double)it;
        double sf = E.scalefactor;
        bool greedy_lane = false;
        // the factor in use: packed Cholesky factor, or the full d x d SVD factor with condmax > 0
        double *Ft = E.usesvd ? E.Rf + (size_t)tile * d * d * 64 : Rt;
        double *F2t = !E.dodr ? nullptr : (E.usesvd ? E.R2f + (size_t)tile * d * d * 64 : E.R2 + (size_t)tile * P * 64);
        const int nf = E.usesvd ? d * d : P;
        if (staypc > 1.0 - E.scalelimit) {
            map_vec(Ft, Ft, lane, nf, [&](double v) { return v / sf; });
            if (E.dodr) {
                double *iCt = E.iC + (size_t)tile * P * 64;
                map_vec(F2t, F2t, lane, nf, [&](double v) { return v / sf; });
                map_vec(iCt, iCt, lane, P, [&](double v) { return v * sf * sf; });
            }
        } else if (staypc < E.scalelimit) {
            map_vec(Ft, Ft, lane, nf, [&](double v) { return v * sf; });
            if (E.dodr) {
                double *iCt = E.iC + (size_t)tile * P * 64;
                map_vec(F2t, F2t, lane, nf, [&](double v) { return v * sf; });
                map_vec(iCt, iCt, lane, P, [&](double v) { return v / sf / sf; });
            }
        } else {
            flags |= ADF_DOCALC;
            greedy_lane = (E.greedy != 0);
        }
        // :83-101 greedy: restart from cmat0 over chain(1:chainind), unit weights
        if (E.greedy != 0 && greedy_lane) {
            flags |= ADF_GREEDY;
            wsum = E.initcmatn;
            for (int e = 0; e < P; ++e) GV(Ct, e) = E.cmat0p[e];
            for (int k = 0; k < d; ++k) GV(mean_t, k) = E.par0[k];
            if (wsum > 0.0) flags |= ADF_STEADY;
            else {
                for (int t = 1; t <= it; ++t) {                       // row list of the one-off batch branch
                    const int slot = t % E.wcap;
                    unsigned long long m = E.wacc[(size_t)tile * E.wcap + slot];
                    if ((m >> lane) & 1ull) { GV(rows, nr) = (uint64_t)(uint32_t)slot | (1ull << 32); ++nr; }
                }
                flags |= ADF_BATCH | ADF_BNOINIT; bstart = 1u;
            }
        }
    } else if (mode & AD_AM) {                                        // MCMC_adapt.F90:105-159
        flags |= ADF_DOCALC;
        if (mode & AD_FIRST) {
            wsum = E.initcmatn;
            for (int e = 0; e < P; ++e) GV(Ct, e) = E.cmat0p[e];
            for (int k = 0; k < d; ++k) GV(mean_t, k) = E.par0[k];
        }
        if (E.adapthist > 1) {
            // AP (:116-136): rows back from chainind until the repeat counts cover adapthist iterations; the oldest
            // row's weight is cut so that the weights sum to adapthist; batch recompute (update = .false.)
            int histsum = (int)curcount;
            int nback = 0;
            uint32_t w = curcount;
            int tt = it - (int)curcount + 1;              // iteration at which the current row was accepted
            GV(rows, 0) = (uint64_t)(uint32_t)(tt % E.wcap) | ((uint64_t)w << 32);
            nback = 1;
            while (histsum < E.adapthist && tt > 1) {
                int t2 = tt - 1, cnt = 1;                 // previous row: accepted at the last set ballot before tt
                while (t2 > 1 && !((E.wacc[(size_t)tile * E.wcap + (t2 % E.wcap)] >> lane) & 1ull)) { --t2; ++cnt; }
                histsum += cnt;
                GV(rows, nback) = (uint64_t)(uint32_t)(t2 % E.wcap) | ((uint64_t)(uint32_t)cnt << 32);
                ++nback; tt = t2;
            }
            {                                             // oldest row's weight: newfreq - histsum + adapthist
                uint64_t e = GV(rows, nback - 1);
                int newfreq = (int)(uint32_t)(e >> 32);
                int wadj = newfreq - histsum + E.adapthist;
                GV(rows, nback - 1) = (e & 0xffffffffull) | ((uint64_t)(uint32_t)wadj << 32);
            }
            // reverse into chain order (oldest first)
            for (int a = 0, b2 = nback - 1; a < b2; ++a, --b2) { uint64_t ta = GV(rows, a); GV(rows, a) = GV(rows, b2); GV(rows, b2) = ta; }
            nr = nback;
            flags |= ADF_BATCH; bstart = (uint32_t)(tt + 1);          // tt: the iteration at which the oldest listed row was accepted
        } else if (wsum > 0.0) {
            flags |= ADF_STEADY;                          // steady state (chainwsum > 0): blocked Welford straight from the ballots
        } else {
            // AM (:138-157), one-off batch branch: rows of chain(lastind:chainind) and their weights, from the accept ballots
            uint32_t w = basecnt;                 // count of the base row when the window started
            uint32_t slot_prev = 0xffffffffu;     // base row lives in basetheta
            for (int t = (int)winstart; t <= it; ++t) {
                const int slot = t % E.wcap;
                unsigned long long m = E.wacc[(size_t)tile * E.wcap + slot];
                if ((m >> lane) & 1ull) {
                    uint32_t wr = (nr == 0) ? (w - lastfreq) : w;
                    GV(rows, nr) = (uint64_t)slot_prev | ((uint64_t)wr << 32);
                    ++nr; slot_prev = (uint32_t)slot; w = 1;
                } else w += 1;
            }
            uint32_t wr = (nr == 0) ? (w - lastfreq) : w;
            GV(rows, nr) = (uint64_t)slot_prev | ((uint64_t)wr << 32);
            ++nr;
            flags |= ADF_BATCH; bstart = winstart;                    // the list's first row is the window's base row
        }
    }
    TIDX(E.ictr, tile, NICTR, I_BSTART, lane) = bstart;
    TIDX(E.ictr, tile, NICTR, I_ADFLAGS, lane) = flags;
    TIDX(E.ictr, tile, NICTR, I_NR, lane) = (uint32_t)nr;
    TIDX(E.scal, tile, NSCAL, S_WSUM, lane) = wsum;
    TIDX(E.scal, tile, NSCAL, S_WNEW, lane) = wsum;
}


// The steady-state form of covmat's Welford update (matutils.F90:283-310), blocked: one block of the upper triangle of cmat stays in
// registers while the window's iterations t0..t1 stream by, so cmat is read and written once per adaptation instead of once per accepted
// row.  Row r of the reference's chain(lastind:chainind) is "the state between two set ballot bits"; its weight (the repeat count) is known
// when the next accept arrives, which is when the row is folded in.  Every component's running mean obeys its own recurrence, so
// recomputing delta = x - mean inside each block repeats the reference's operations exactly (deltas, o = delta_u delta_v, C += f1 (f2 o -
// C), means).  The window's base row (basetheta), its count at window start and the amount (lastfreq) taken off the first folded weight are
// the AM branch's (MCMC_adapt.F90:140-147); the greedy restart (:91) has unit weights and no base row. Blocks of TD = 10 (the BASELINE
// dimensions 10, 20, 50 are whole numbers of them): a DIAGONAL block is its upper triangle, 55 elements, an off-diagonal one 100 -- fewer,
// larger blocks repeat the per-fold overhead (the three divisions, the deltas, the row's loads) less often than the 8 x 8 cover of round 3
// did (tools/variants/README.md).  DIAG: launched with two waves per SIMD; the off-diagonal form holds 100 accumulators and runs one wave
// per SIMD (its hundred independent chains keep the VALU busy without a second wave). The walk is in lockstep and a fold runs under the
// exec mask of whoever accepted at that iteration; decoupling walk and folds by a per-lane FIFO in LDS (fewer, fuller rounds) was built in
// round 6, is bit-equal, and LOSES at the bench's acceptance rates: tools/variants/mcx_cov_fifo.hpp. Grid: 8 * ceil(ntiles / 8) * nblk
// workgroups of one wave; workgroup w runs on XCD w % 8, so tile = (w / 8 / nblk) * 8 + w % 8, block = (w / 8) % nblk keeps a tile's blocks
// on one XCD and next to each other in time.
constexpr int TD = 10;
template <bool DIAG>
MCX_DEV void covmat_window_td(const EngineDev &E, int it, int mode, int nblk)
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
    bool have = act && !unit;
    uint32_t cnt = count0, adj = adj0;
    auto fold = [&](bool on, double w3) {
        if (on) {
            const double f1 = w3 / (W + w3 - 1.0), f2 = W / (W + w3), f3 = w3 / (W + w3);
#pragma unroll
            for (int u = 0; u < TD; ++u) { xa[u] = xa[u] - ma[u]; if (!DIAG) xb[u] = xb[u] - mb[u]; }
#pragma unroll
            for (int u = 0; u < TD; ++u)
#pragma unroll
                for (int v = (DIAG ? u : 0); v < TD; ++v) {
                    double o = xa[u] * (DIAG ? xa[v] : xb[v]);
                    C[u][v] = C[u][v] + f1 * (f2 * o - C[u][v]);
                }
#pragma unroll
            for (int u = 0; u < TD; ++u) { ma[u] = ma[u] + f3 * xa[u]; if (!DIAG) mb[u] = mb[u] + f3 * xb[u]; }
            W = w3 + W;
        }
    };
    for (int tc = t0; tc <= t1; tc += 64) {
        const int tl = tc + lane;
        const unsigned long long mine = (tl <= t1) ? wacc_t[tl % E.wcap] : 0ull;
        const int nq = (t1 - tc + 1) < 64 ? (t1 - tc + 1) : 64;
        for (int q = 0; q < nq; ++q) {
            const int t = tc + q, slot = t % E.wcap;
            const unsigned long long m = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(mine >> 32), q) << 32)
                                         | (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)mine, q);
            const bool inwin = act && (t >= t0lane);
            const bool acc = inwin && ((m >> lane) & 1ull);
            if (__any(acc)) {
                double xan[TD], xbn[NBV];
                if (acc) {
                    const size_t so = (size_t)slot * (size_t)E.hs * 64;
#pragma unroll
                    for (int u = 0; u < TD; ++u) {
                        xan[u] = hist_t[so + (size_t)((a0 + u < d) ? a0 + u : d - 1) * 64 + lane];
                        if (!DIAG) xbn[u] = hist_t[so + (size_t)((b0 + u < d) ? b0 + u : d - 1) * 64 + lane];
                    }
                }
                const bool fl = acc && have;
                if (__any(fl)) fold(fl, unit ? 1.0 : (double)(cnt - adj));
                if (acc) {
#pragma unroll
                    for (int u = 0; u < TD; ++u) { xa[u] = xan[u]; if (!DIAG) xb[u] = xbn[u]; }
                    if (have) adj = 0;
                    have = true; cnt = 1;
                }
            }
            if (inwin && !acc) cnt += 1;
        }
    }
    if (__any(have)) fold(have, unit ? 1.0 : (double)(cnt - adj));
#pragma unroll
    for (int u = 0; u < TD; ++u) {
        const int a = a0 + u;
#pragma unroll
        for (int v = (DIAG ? u : 0); v < TD; ++v) { const int b = b0 + v; if (act && a < d && b < d) GV(Ct, pidx(a, b, d)) = C[u][v]; }
        if (DIAG && act && a < d) GV(mnew_t, a) = ma[u];     // the other blocks still need the old means
    }
    if (DIAG && a0 == 0 && act) TIDX(E.scal, tile, NSCAL, S_WNEW, lane) = W;
}
// covmat's batch branch (matutils.F90:311-338: weighted mean first, then sum_r (x_ri - m_i) ((x_rj - m_j) w_r), divided by wsum - 1)
// in the same blocks: the rows of the lane's list (adapt_pre_kernel: ring slot | weight << 32) are the states between set ballot
// bits from I_BSTART on, so the block walks the window's iterations in lockstep like the steady form -- a row's loads are whole
// 512-byte segments whichever lanes want them -- and takes each row's WEIGHT from the list when the next accept closes it.  Two
// walks (means, then products); element for element the operations of covmat_rows' batch branch, which visits cmat once per ROW
// (84 ms for the first adaptation of 131072 chains at npar = 50, and every adaptation of an AP run).
template <bool DIAG>
MCX_DEV void covmat_batch_td(const EngineDev &E, int it, int nblk)
{
    const int lane = threadIdx.x, d = E.d, P = E.P;
    const int w = blockIdx.x, j = w >> 3;
    const int tile = (j / nblk) * 8 + (w & 7);
    int blk = j % nblk;
    if (tile >= E.ntiles) return;
    const uint32_t flags = TIDX(E.ictr, tile, NICTR, I_ADFLAGS, lane);
    const bool act = (flags & ADF_BATCH) != 0, noinit = (flags & ADF_BNOINIT) != 0;
    if (!__any(act)) return;
    const int nb = (d + TD - 1) / TD;
    int a0 = 0, b0 = 0;
    if (DIAG) { a0 = b0 = blk * TD; }
    else { int ar = 0; while (blk >= nb - 1 - ar) { blk -= nb - 1 - ar; ++ar; } a0 = ar * TD; b0 = (ar + 1 + blk) * TD; }
    double *Ct = E.cmat + (size_t)tile * P * 64;
    const double *base_t = E.basetheta + (size_t)tile * d * 64;
    double *mnew_t = E.cand + (size_t)tile * d * 64;
    const uint64_t *rows = E.rowlist + (size_t)tile * (E.wcap + 1) * 64;
    const int t0lane = (int)TIDX(E.ictr, tile, NICTR, I_BSTART, lane), t1 = it;
    int t0 = act ? t0lane : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int other = __shfl_xor(t0, o); t0 = other < t0 ? other : t0; }
    const double *hist_t = E.hist + (size_t)tile * E.wcap * (size_t)E.hs * 64;
    const unsigned long long *wacc_t = (const unsigned long long *)E.wacc + (size_t)tile * E.wcap;
    constexpr int NBV = DIAG ? 1 : TD;
    double C[TD][TD], ma[TD], xa[TD], mb[NBV], xb[NBV];
    // one walk over the window: fold(on, weight) closes the row in xa / xb for the lanes `on`
    auto walk = [&](auto &&fold) {
        bool have = act && !noinit;
        int idx = have ? 0 : -1;                          // the list entry of the open row
        // the list's first row dates from before the window: the base row, or a ring slot of the lane's own
        if (have) {
            const uint32_t slot = (uint32_t)GV(rows, 0);
            const bool isbase = (slot == 0xffffffffu);
            const size_t so = isbase ? 0 : (size_t)slot * (size_t)E.hs * 64;
#pragma unroll
            for (int u = 0; u < TD; ++u) {
                const int a = (a0 + u < d) ? a0 + u : d - 1;
                xa[u] = isbase ? GV(base_t, a) : hist_t[so + (size_t)a * 64 + lane];
                if (!DIAG) { const int b = (b0 + u < d) ? b0 + u : d - 1; xb[u] = isbase ? GV(base_t,
                    b) : hist_t[so + (size_t)b * 64 + lane]; }
            }
        }
        for (int tc = t0; tc <= t1; tc += 64) {
            const int tl = tc + lane;
            const unsigned long long mine = (tl <= t1) ? wacc_t[tl % E.wcap] : 0ull;
            const int nq = (t1 - tc + 1) < 64 ? (t1 - tc + 1) : 64;
            for (int q = 0; q < nq; ++q) {
                const int t = tc + q, slot = t % E.wcap;
                const unsigned long long m = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(mine >> 32), q) << 32)
                                             | (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)mine, q);
                const bool acc = act && (t >= t0lane) && ((m >> lane) & 1ull);
                if (__any(acc)) {
                    double xan[TD], xbn[NBV];
                    if (acc) {
                        const size_t so = (size_t)slot * (size_t)E.hs * 64;
#pragma unroll
                        for (int u = 0; u < TD; ++u) {
                            xan[u] = hist_t[so + (size_t)((a0 + u < d) ? a0 + u : d - 1) * 64 + lane];
                            if (!DIAG) xbn[u] = hist_t[so + (size_t)((b0 + u < d) ? b0 + u : d - 1) * 64 + lane];
                        }
                    }
                    const bool fl = acc && have;
                    if (__any(fl)) fold(fl, fl ? (double)(uint32_t)(GV(rows, idx) >> 32) : 0.0);
                    if (acc) {
#pragma unroll
                        for (int u = 0; u < TD; ++u) { xa[u] = xan[u]; if (!DIAG) xb[u] = xbn[u]; }
                        have = true; idx += 1;
                    }
                }
            }
        }
        if (__any(have)) fold(have, have ? (double)(uint32_t)(GV(rows, idx) >> 32) : 0.0);
    };
    // ---- xmean2 = sum_r x_r w_r / sum_r w_r (rows in list order)
    double wsum2 = 0.0;
#pragma unroll
    for (int u = 0; u < TD; ++u) { ma[u] = 0.0; if (!DIAG) mb[u] = 0.0; }
    walk([&](bool on, double w3) {
        if (on) {
#pragma unroll
            for (int u = 0; u < TD; ++u) { ma[u] = ma[u] + xa[u] * w3; if (!DIAG) mb[u] = mb[u] + xb[u] * w3; }
            wsum2 = wsum2 + w3;
        }
    });
#pragma unroll
    for (int u = 0; u < TD; ++u) { ma[u] = ma[u] / wsum2; if (!DIAG) mb[u] = mb[u] / wsum2; }
    // ---- cmat(j,k), j <= k: sum_r (x_rk - m_k) ((x_rj - m_j) w_r), then / (wsum - 1)
#pragma unroll
    for (int u = 0; u < TD; ++u)
#pragma unroll
        for (int v = 0; v < TD; ++v) C[u][v] = 0.0;
    walk([&](bool on, double w3) {
        if (on) {
            double da[TD], db[NBV];
#pragma unroll
            for (int u = 0; u < TD; ++u) { da[u] = xa[u] - ma[u]; if (!DIAG) db[u] = xb[u] - mb[u]; }
#pragma unroll
            for (int u = 0; u < TD; ++u) {
                const double xw = da[u] * w3;
#pragma unroll
                for (int v = (DIAG ? u : 0); v < TD; ++v) C[u][v] = C[u][v] + (DIAG ? da[v] : db[v]) * xw;
            }
        }
    });
#pragma unroll
    for (int u = 0; u < TD; ++u) {
        const int a = a0 + u;
#pragma unroll
        for (int v = (DIAG ? u : 0); v < TD; ++v) { const int b = b0 + v; if (act && a < d && b < d) GV(Ct, pidx(a, b,
            d)) = C[u][v] / (wsum2 - 1.0); }
        if (DIAG && act && a < d) GV(mnew_t, a) = ma[u];
    }
    if (DIAG && a0 == 0 && act) TIDX(E.scal, tile, NSCAL, S_WNEW, lane) = wsum2;
}
__global__ __launch_bounds__(64, 2) void adapt_covb_diag_kernel(EngineDev E, int it, int nblk) { covmat_batch_td<true>(E, it, nblk); }
__global__ __launch_bounds__(64, 1) void adapt_covb_off_kernel(EngineDev E, int it, int nblk) { covmat_batch_td<false>(E, it, nblk); }
__global__ __launch_bounds__(64, 2) void adapt_cov_diag_kernel(EngineDev E, int it, int mode, int nblk) { covmat_window_td<true>(E, it,
    mode, nblk); }
__global__ __launch_bounds__(64, 1) void adapt_cov_off_kernel(EngineDev E, int it, int mode, int nblk) { covmat_window_td<false>(E, it,
    mode, nblk); }

// phase 0: the whole tick.  With the blocked SVD (large npar, below) the tick is cut around the factorisation:
// phase 1 = everything up to and including the symmetric matrix in Gw (and the per-chain `need` flags),
// phase 2 = everything after the SVD (which has left the singular vectors in Vw and the singular values in cs).
// SVD: the instance with the SVD branches of MCMC_calculate_R (condmax > 0, scam); the Cholesky instance keeps to 256 registers
// (two waves per SIMD: its sweeps wait on loads)
#ifndef MCX_POST_WAVES
#define MCX_POST_WAVES 2
#endif
// XG (npar > 320: one npar-vector per lane no longer fits a CU's LDS): the work vector in the tile's global scratch (EngineDev::xscr) -- a
// compile-time choice, so that neither form uses flat accesses.  Slower; any npar.
template <bool SVD, bool XG = false>
__global__ __launch_bounds__(64, SVD ? 1 : MCX_POST_WAVES) void adapt_post_kernel(EngineDev E, int it, int mode, int phase, uint8_t *need,
    int batch_done)
{
    extern __shared__ double Xlds[];
    const int lane = threadIdx.x, tile = blockIdx.x, d = E.d, P = E.P;
    double *X = XG ? E.xscr + (size_t)tile * 2 * d * 64 : Xlds;
    double *Rt = E.R + (size_t)tile * P * 64;
    double *Ct = E.cmat + (size_t)tile * P * 64;
    double *Tt = E.Rtmp + (size_t)tile * P * 64;
    double *mean_t = E.mean + (size_t)tile * d * 64;
    double *base_t = E.basetheta + (size_t)tile * d * 64;
    double *theta_t = E.theta + (size_t)tile * d * 64;
    // the blocked update's new means; then scratch (xmean2 of the batch branch)
    double *m2_t = E.cand + (size_t)tile * d * 64;
    uint64_t *rows = E.rowlist + (size_t)tile * (E.wcap + 1) * 64;
    uint32_t curcount = TIDX(E.ictr, tile, NICTR, I_CURCOUNT, lane);
    uint32_t lastfreq = TIDX(E.ictr, tile, NICTR, I_LASTFREQ, lane);
    uint32_t basecnt = TIDX(E.ictr, tile, NICTR, I_BASECNT, lane);
    uint32_t winstart = TIDX(E.ictr, tile, NICTR, I_WINSTART, lane);
    const uint32_t flags = TIDX(E.ictr, tile, NICTR, I_ADFLAGS, lane);
    const int nr = (int)TIDX(E.ictr, tile, NICTR, I_NR, lane);
    const bool docalc = (flags & ADF_DOCALC) != 0, greedy_lane = (flags & ADF_GREEDY) != 0;
    // lanes whose covariance and mean the blocked kernels have already updated: the steady Welford form, and (batch_done: the
    // host launched adapt_covb_*) the batch branch over the row list
    const bool steady = (flags & ADF_STEADY) != 0 || (batch_done != 0 && (flags & ADF_BATCH) != 0);
    double wsum = TIDX(E.scal, tile, NSCAL, (steady && phase != 2) ? S_WNEW : S_WSUM, lane);
    if (phase != 2) {
    if (steady) copy_vec(mean_t, m2_t, nullptr, lane, d);

    if (mode & AD_BURN) {
        if (E.greedy != 0) {
            covmat_rows(E, tile, lane, rows, nr, greedy_lane && !steady, true, Ct, mean_t, base_t, m2_t, wsum, X);
            if (greedy_lane) lastfreq = curcount;
        }
        if (docalc) {
            // lastind = chainind: the covariance window restarts at the current row (lastfreq only touched by greedy)
            copy_vec(base_t, theta_t, nullptr, lane, d);
            basecnt = curcount; winstart = (uint32_t)(it + 1);
        }
    } else if (mode & AD_AM) {
        if (E.adapthist > 1) {
            covmat_rows(E, tile, lane, rows, nr, !steady, false, Ct, mean_t, base_t, m2_t, wsum, X);
        } else {
            covmat_rows(E, tile, lane, rows, nr, !steady, true, Ct, mean_t, base_t, m2_t, wsum, X);
            // lastfreq = count of the current row; lastind = chainind -> window restarts here
            lastfreq = curcount;
            copy_vec(base_t, theta_t, nullptr, lane, d);
            basecnt = curcount; winstart = (uint32_t)(it + 1);
        }
    }

    TIDX(E.ictr, tile, NICTR, I_LASTFREQ, lane) = lastfreq;
    TIDX(E.ictr, tile, NICTR, I_BASECNT, lane) = basecnt;
    TIDX(E.ictr, tile, NICTR, I_WINSTART, lane) = winstart;
    TIDX(E.scal, tile, NSCAL, S_WSUM, lane) = wsum;
    }
    if (SVD) {
        // MCMC_calculate_R, SVD branches (MCMC_adapt.F90:189-209): covtor_svd / scam_svd (matutils.F90:378-453, 583-653)
        double *Gt = E.Gw + (size_t)tile * d * d * 64, *Vt = E.Vw + (size_t)tile * d * d * 64;
        double *Rft = E.Rf + (size_t)tile * d * d * 64;
        double *sv_t = E.cs + (size_t)tile * 2 * d * 64;
        if (phase == 1) need[tile * 64 + lane] = docalc ? 1 : 0;
        if (__any(docalc)) {
            if (phase != 2 && docalc) for (int j = 0; j < d; ++j) for (int i = 0; i < d; ++i)
                GV(Gt, (size_t)j * d + i) = (i <= j) ? GV(Ct, pidx(i, j, d)) : GV(Ct, pidx(j, i, d));
            if (phase == 1) return;
            if (phase == 0) symsvd_dev(Gt, Vt, sv_t, lane, d, docalc);
            if (docalc) {
                int info = 0;
                const double s0 = GV(sv_t, 0);
                if (s0 == 0.0) { info = d; TIDX(E.ictr, tile, NICTR, I_STATUS, lane) |= ST_CHOL_FAIL; }
                else {
                    const double tol = s0 / E.condmax;
                    bool floored = false;
                    if (GV(sv_t, d - 1) <= tol) {
                        floored = true;
                        for (int i = 0; i < d; ++i) if (GV(sv_t, i) < tol) GV(sv_t, i) = tol;
                    }
                    if (E.doscam) {                                   // R = U, qcovstd = sqrt(s)
                        copy_vec(Rft, Vt, nullptr, lane, d * d);
                        double *std_t = E.qstd + (size_t)tile * d * 64;
                        for (int i = 0; i < d; ++i) GV(std_t, i) = sqrt(GV(sv_t, i));
                    } else {                                          // R0 = U diag(sqrt(s)); R = R0*2.4/sqrt(d)
                        for (int i = 0; i < d; ++i) {
                            const double sq = sqrt(GV(sv_t, i));
                            for (int k = 0; k < d; ++k) GV(Vt, (size_t)i * d + k) = sq * GV(Vt, (size_t)i * d + k);
                        }
                        if (floored) {                                // cmat = matmul(R0, transpose(R0))
                            for (int j = 0; j < d; ++j)
                                for (int i = 0; i <= j; ++i) {
                                    double acc = 0.0;
                                    for (int k = 0; k < d; ++k) acc = dfma(GV(Vt, (size_t)k * d + i), GV(Vt, (size_t)k * d + j), acc);
                                    GV(Ct, pidx(i, j, d)) = acc;
                                }
                        }
                        const double sqd = sqrt((double)d);
                        map_vec(Rft, Vt, lane, d * d, [&](double v) { return v * 2.4 / sqd; });
                        if (E.dodr) {                                 // iC = dpotri('u', R): on R's upper triangle; R2 = R/drscale
                            double *iCt = E.iC + (size_t)tile * P * 64, *R2ft = E.R2f + (size_t)tile * d * d * 64;
                            for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i) GV(iCt, pidx(i, j, d)) = GV(Rft, (size_t)j * d + i);
                            int info2 = potri_packed(iCt, lane, d, true, X);
                            if (info2 != 0) TIDX(E.ictr, tile, NICTR, I_STATUS, lane) |= ST_POTRI_FAIL;
                            map_vec(R2ft, Rft, lane, d * d, [&](double v) { return v / E.drscale; });
                        }
                    }
                }
                TIDX(E.ictr, tile, NICTR, I_INFO, lane) = (uint32_t)info;
            }
        }
    } else if (phase != 3 && __any(docalc)) {           // (phase 3: tile_factor_kernel has the factorisation)
        int info = calculate_R(Ct, Tt, Rt, lane, d, P, docalc, X);
        if (docalc) {
            TIDX(E.ictr, tile, NICTR, I_INFO, lane) = (uint32_t)info;
            if (info != 0) TIDX(E.ictr, tile, NICTR, I_STATUS, lane) |= ST_CHOL_FAIL;   // warning, old R kept (:168-171)
        }
        if (E.dodr) {                                       // iC = dpotri(R), R2 = R/drscale (:216-225)
            const bool ok = docalc && info == 0;
            double *R2t = E.R2 + (size_t)tile * P * 64, *iCt = E.iC + (size_t)tile * P * 64;
            if (ok) copy_vec(iCt, Rt, nullptr, lane, P);
            int info2 = potri_packed(iCt, lane, d, ok, X);
            if (ok && info2 != 0) TIDX(E.ictr, tile, NICTR, I_STATUS, lane) |= ST_POTRI_FAIL;  // the reference stops
            if (ok) map_vec(R2t, Rt, lane, P, [&](double v) { return v / E.drscale; });
        }
    }
}

} // namespace mcx
