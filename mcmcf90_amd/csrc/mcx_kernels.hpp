// mcx_kernels.hpp -- HIP kernels of the adaptive-Metropolis engine (gfx950, wave64).
//
// Execution model: one lane = one chain, one 64-lane wave (= one workgroup) = one
// "tile" of 64 chains.  All per-chain arrays in HBM are tile-interleaved,
//     element k of chain (tile, lane)  ->  base[(tile*K + k)*64 + lane],
// i.e. parameter-major inside a tile, so every wave access is one contiguous
// 512-byte segment and a tile's whole Cholesky factor is one sequential stream.
// One per-lane d-vector lives in a VGPR array with compile-time indices; the O(d^2) loops are
// "runtime row index, unrolled column index" with the row's elements loaded 8 at a time, so a
// wave always has several independent 512-byte loads in flight.
#pragma once
#include "mcx_device.hpp"

namespace mcx {

enum { TGT_GAUSS = 0, TGT_BANANA = 1, TGT_EXPDATA = 2 };
enum { M_DRAM = 0, M_RAM = 1 };

// per-chain scalar slots (doubles)
enum { S_SS1 = 0, S_PRI1, S_SIGMA2, S_ALPHA12, S_SAVEDY, S_WSUM, NSCAL };
// per-chain integer slots (u32)
enum { I_SAVED = 0, I_STAYED, I_BNDSTAYED, I_DRACC, I_DRTRIES, I_CHAININD, I_CURCOUNT, I_STATUS,
       I_LASTFREQ, I_BASECNT, I_WINSTART, I_INFO, NICTR };

// status bits
enum { ST_RAM_DOWNDATE_FAIL = 1, ST_CHOL_FAIL = 2 };

struct DevTarget {
    int kind;
    const double *mu, *lam;     // gauss: mean[d], precision row-major [d*d]
    double b;                   // banana
    int ndata;                  // expdata
    const double *x, *y;
    const double *lo, *hi;      // box bounds or nullptr
    const double *pmu, *psig;   // Gaussian priors or nullptr
};

struct EngineDev {
    int d, P, ntiles;
    int method, dodr, updatesigma, doadapt, doburnin, burnintime;
    double gam_shape;           // N0/2 + nobs/2           (MCMC_DRAM.F90:201)
    double N0S02;               // N0*S02
    double alphatarget, drscale, scalelimit, scalefactor;
    DevTarget tgt;
    // state, tile-interleaved
    double *theta, *cand, *zs, *cs, *scal, *R, *R2, *iC, *Rtmp;   // cand/zs [d], cs [2d]: per-chain scratch vectors
    double *cmat, *mean, *basetheta;
    const double *cmat0p, *par0;    // packed upper cmat0 [P], par0 [d] (shared by all chains)
    uint32_t *ictr;
    uint64_t *rngn;
    uint32_t k0, chain_id0;
    // history ring: slot = it % wcap; hist[(tile*wcap + slot)*(d+1) + k][lane]
    int wcap, record_s2;
    double *hist, *s2hist;
    uint64_t *wacc;             // [tile*wcap + slot]
    uint64_t *accmask;          // [(it-1)*ntiles + tile] or nullptr
    uint64_t *rowlist;          // [(tile*(wcap+1) + r)*64 + lane]  (slot | weight<<32)
};

#define TIDX(base, tile, K, k, lane) ((base)[((size_t)(tile) * (size_t)(K) + (size_t)(k)) * 64 + (lane)])

// Packed upper triangle, ROW-major: element (i,j), i <= j, sits at rowstart(i) + (j - i).
// Every sweep of the factor (proposal, update, downdate) walks whole rows, forwards or backwards,
// so a tile's factor is one sequential HBM stream of 512-byte wave segments.
MCX_DEV int rowstart(int i, int d) { return i * d - (i * (i - 1)) / 2; }
MCX_DEV int pidx(int i, int j, int d) { return rowstart(i, d) + (j - i); }

constexpr int CH = 8;     // row elements loaded per batch (8 x 512 B in flight per wave and batch)

// Visit row i of a packed factor: the elements j0..j0+CH-1 of each chunk that reaches the row are
// loaded together (clamped addresses, unconditional), then f(j, r, valid, interior) runs per element
// with a compile-time j.  interior == true: the whole chunk is strictly right of the diagonal and
// inside d, so the body needs no predication.
template <int D, typename F>
MCX_DEV void sweep_row(int i, int d, const double *rowp, F &&f)
{
    constexpr int NCH = (D + CH - 1) / CH;
    const int m = d - 1 - i;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int j0 = c * CH;
        if (j0 + CH - 1 >= i && j0 < d) {
            double r[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                int o = j0 + u - i; o = o < 0 ? 0 : o; o = o > m ? m : o;
                r[u] = rowp[(size_t)o * 64];
            }
            if (j0 > i && j0 + CH <= d) {
#pragma unroll
                for (int u = 0; u < CH; ++u) if (j0 + u < D) f(j0 + u, r[u], true, true);
            } else {
#pragma unroll
                for (int u = 0; u < CH; ++u) if (j0 + u < D) f(j0 + u, r[u], (j0 + u >= i) && (j0 + u < d), false);
            }
        }
    }
}

// ---------------------------------------------------------------- targets (user ssfunction / priorfun / checkbounds)
// th[] holds the candidate on entry and (th - mu) on exit for the Gaussian target; cand_t is the
// same candidate in global scratch (element stride 64), used for the few runtime-indexed reads.
template <int D>
MCX_DEV double target_ss(const DevTarget &t, int d, double (&th)[D], const double *cand_t)
{
    double ss = 0.0;
    if (t.kind == TGT_GAUSS) {
        // ss = (th-mu)' Lam (th-mu): y_i = sum_j lam(i,j) v_j ascending (fma chain), ss = sum_i y_i v_i (fma chain)
#pragma unroll
        for (int j = 0; j < D; ++j) if (j < d) th[j] = th[j] - t.mu[j];
        for (int i = 0; i < d; ++i) {
            const double *__restrict__ row = t.lam + (size_t)i * d;
            double y = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                if (j < d) { if (j == 0) y = row[0] * th[0]; else y = dfma(row[j], th[j], y); }
            }
            double vi = cand_t[(size_t)i * 64] - t.mu[i];
            if (i == 0) ss = y * vi; else ss = dfma(y, vi, ss);
        }
    } else if (t.kind == TGT_BANANA) {
        double t1 = th[0] * th[0];
        double q = dfma(t.b, t1, th[D > 1 ? 1 : 0]) - 100.0 * t.b;
        ss = dfma(q, q, t1 / 100.0);
#pragma unroll
        for (int k = 2; k < D; ++k) if (k < d) ss = dfma(th[k], th[k], ss);
    } else {
        for (int i = 0; i < t.ndata; ++i) {
            double r = t.y[i] - th[0] * d_exp(-(th[D > 1 ? 1 : 0] * t.x[i]));
            ss = dfma(r, r, ss);
        }
    }
    return ss;
}

template <int D>
MCX_DEV double target_prior(const DevTarget &t, int d, const double (&th)[D])
{
    double p = 0.0;
    if (t.pmu) {
#pragma unroll
        for (int i = 0; i < D; ++i)
            if (i < d) { double sg = t.psig[i]; if (sg > 0.0) { double q = (th[i] - t.pmu[i]) / sg; p = p + q * q; } }
    }
    return p;
}

template <int D>
MCX_DEV bool target_inbounds(const DevTarget &t, int d, const double (&th)[D])
{
    bool ok = true;
#pragma unroll
    for (int i = 0; i < D; ++i)
        if (i < d) {
            if (t.lo) ok = ok && (th[i] > t.lo[i]);
            if (t.hi) ok = ok && (th[i] < t.hi[i]);
        }
    return ok;
}

// ---------------------------------------------------------------- normals (mcmcrand.F90:60-83,166-190)
// Each lane appends accepted polar pairs to its own column of zs (global scratch, element stride 64)
// until it has d deviates; the wave loops until every participating lane is done.  The cached
// second deviate of normal_bm is honoured and left behind when d is odd.
MCX_DEV void gen_normals(Rng &g, double *zs_t, int d, bool participate)
{
    int k = 0;
    if (participate && g.saved && d > 0) { zs_t[0] = g.saved_y; g.saved = 0; k = 1; }
    bool need = participate && (k < d);
    while (__any(need)) {
        if (need) {
            double a, b;
            if (polar_try(g, a, b)) {
                zs_t[(size_t)k * 64] = a; ++k;
                if (k < d) { zs_t[(size_t)k * 64] = b; ++k; }
                else { g.saved_y = b; g.saved = 1; }
            }
            need = (k < d);
        }
    }
}

// ---------------------------------------------------------------- proposal: P = R'z  (MCMC_DRAM.F90:20-31)
// dtrmv('U','T','N') (matutils.F90:108-109) in netlib accumulation order: p_j = z_j R(j,j), then
// + R(i,j) z_i for i = j-1..0 as an fma chain -- which is what a sweep over rows i = d-1..0 produces.
template <int D>
MCX_DEV void trmv_rows(const double *Rt, const double *zs_t, int d, double (&P)[D])
{
    for (int i = d - 1; i >= 0; --i) {
        const double zi = zs_t[(size_t)i * 64];
        const double *rowp = Rt + (size_t)rowstart(i, d) * 64;
        sweep_row<D>(i, d, rowp, [&](int j, double r, bool valid, bool interior) {
            if (interior) P[j] = dfma(r, zi, P[j]);
            else {
                double nv = (j == i) ? zi * r : dfma(r, zi, P[j]);
                P[j] = valid ? nv : P[j];
            }
        });
    }
}

// ---------------------------------------------------------------- RAM rank-1 adaptation (MCMC_run_ram.F90:104-179)
// a >= 0: cholupdate = DCHUD (dchud.f:122-139); a < 0: choldowndate = DCHDD (dchdd.f:141-179), both
// restated row by row (same operations on every element, in the same order per column).
// X is the one per-lane register array; cs_t is global scratch for the downdate's rotations.
template <int D>
MCX_DEV void ram_update(double *Rt, const double *zs_t, double *cs_t, int d, double a, bool act,
                        double (&X)[D], uint32_t &status)
{
    double su = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) if (k < d) { X[k] = zs_t[(size_t)k * 64]; }
#pragma unroll
    for (int k = 0; k < D; ++k) if (k < d) { su = su + X[k] * X[k]; }
    const bool up = act && (a >= 0.0);
    const bool down = act && !(a >= 0.0);
    if (__any(up)) {
        if (up) {
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < d) X[k] = X[k] / su * a;          // x = u/sum(u**2) * a
            double xdiag = X[0];
            for (int i = 0; i < d; ++i) {
                double *rowp = Rt + (size_t)rowstart(i, d) * 64;
                double r, c, s;
                d_rotg(rowp[0], xdiag, r, c, s);
                rowp[0] = r;
                sweep_row<D>(i, d, rowp, [&](int j, double rij, bool valid, bool interior) {
                    if (interior) {
                        double t = c * rij + s * X[j];
                        X[j] = c * X[j] - s * rij;
                        rowp[(size_t)(j - i) * 64] = t;
                        if ((j % CH) == 0) xdiag = (j == i + 1) ? X[j] : xdiag;
                    } else {
                        const bool off = valid && (j > i);
                        double t = c * rij + s * X[j];
                        double nx = c * X[j] - s * rij;
                        X[j] = off ? nx : X[j];
                        if (off) rowp[(size_t)(j - i) * 64] = t;
                        xdiag = (off && j == i + 1) ? X[j] : xdiag;
                    }
                });
            }
        }
    }
    if (__any(down)) {
        if (down) {
            // solve R'a = x, x = -u/sum(u**2)*a (dchdd.f:141-148); slot j of X is the running dot of column j
            // until row j, then the solution s_j
            const double fac = a;
            double accdiag = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < d) X[k] = 0.0;
            for (int i = 0; i < d; ++i) {
                const double *rowp = Rt + (size_t)rowstart(i, d) * 64;
                double xi = -(zs_t[(size_t)i * 64] / su * fac);
                double si = xi - accdiag;
                si = si / rowp[0];
                sweep_row<D>(i, d, rowp, [&](int j, double rij, bool valid, bool interior) {
                    if (interior) {
                        X[j] = dfma(rij, si, X[j]);
                        if ((j % CH) == 0) accdiag = (j == i + 1) ? X[j] : accdiag;
                    } else {
                        const bool off = valid && (j > i);
                        double na = dfma(rij, si, X[j]);
                        X[j] = off ? na : ((valid && j == i) ? si : X[j]);
                        accdiag = (off && j == i + 1) ? X[j] : accdiag;
                    }
                });
            }
            // norm = dnrm2(p, s), classic scale/ssq form (dchdd.f:149)
            double norm;
            if (d == 1) norm = fabs(X[0]);
            else {
                double scale = 0.0, ssq = 1.0;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    if (k < d && X[k] != 0.0) {
                        double ax = fabs(X[k]);
                        if (scale < ax) { double q = scale / ax; ssq = 1.0 + ssq * (q * q); scale = ax; }
                        else { double q = ax / scale; ssq = ssq + q * q; }
                    }
                }
                norm = scale * sqrt(ssq);
            }
            if (!(norm < 1.0)) {
                status |= ST_RAM_DOWNDATE_FAIL;      // INFO = -1: R untouched (the reference stops here)
            } else {
                double alpha = sqrt(1.0 - norm * norm);
#pragma unroll
                for (int k = D - 1; k >= 0; --k) {   // dchdd.f:158-167
                    if (k < d) {
                        double scale = alpha + fabs(X[k]);
                        double aa = alpha / scale, bb = X[k] / scale;
                        double nn = sqrt(aa * aa + bb * bb);
                        cs_t[(size_t)(2 * k) * 64] = aa / nn;
                        cs_t[(size_t)(2 * k + 1) * 64] = bb / nn;
                        alpha = scale * nn;
                    }
                }
#pragma unroll
                for (int k = 0; k < D; ++k) if (k < d) X[k] = 0.0;          // xx of every column
                double cn = cs_t[(size_t)(2 * (d - 1)) * 64], sn = cs_t[(size_t)(2 * (d - 1) + 1) * 64];
                for (int i = d - 1; i >= 0; --i) {   // dchdd.f:171-179, rows d-1..0
                    double *rowp = Rt + (size_t)rowstart(i, d) * 64;
                    const double ci = cn, si = sn;
                    if (i > 0) { cn = cs_t[(size_t)(2 * (i - 1)) * 64]; sn = cs_t[(size_t)(2 * (i - 1) + 1) * 64]; }
                    sweep_row<D>(i, d, rowp, [&](int j, double rij, bool valid, bool interior) {
                        double t = ci * X[j] + si * rij;
                        double nr = ci * rij - si * X[j];
                        if (interior) { rowp[(size_t)(j - i) * 64] = nr; X[j] = t; }
                        else { if (valid) rowp[(size_t)(j - i) * 64] = nr; X[j] = valid ? t : X[j]; }
                    });
                }
            }
        }
    }
}

// ---------------------------------------------------------------- the step kernel
// Iterations it0..it1 (absolute simuind) of MCMC_run (MCMC_run.F90:41-107, no DR stage here)
// or MCMC_run_ram (MCMC_run_ram.F90:45-81) for one tile of 64 chains.
template <int D>
__global__ __launch_bounds__(64) void step_kernel(EngineDev E, int it0, int it1, const double *__restrict__ ramscale)
{
    const int lane = threadIdx.x, tile = blockIdx.x, d = E.d;
    const double *theta_t = E.theta + (size_t)tile * d * 64 + lane;
    double *theta_w = E.theta + (size_t)tile * d * 64 + lane;
    double *cand_t = E.cand + (size_t)tile * d * 64 + lane;
    double *zs_t = E.zs + (size_t)tile * d * 64 + lane;
    double *cs_t = E.cs + (size_t)tile * 2 * d * 64 + lane;
    double *Rt = E.R + (size_t)tile * E.P * 64 + lane;

    Rng g;
    g.k0 = E.k0; g.k1 = E.chain_id0 + (uint32_t)(tile * 64 + lane);
    g.n = TIDX(E.rngn, tile, 1, 0, lane); g.cblk = 0; g.c2 = 0; g.c3 = 0;
    g.saved = (int)TIDX(E.ictr, tile, NICTR, I_SAVED, lane);
    g.saved_y = TIDX(E.scal, tile, NSCAL, S_SAVEDY, lane);
    double ss1 = TIDX(E.scal, tile, NSCAL, S_SS1, lane), pri1 = TIDX(E.scal, tile, NSCAL, S_PRI1, lane);
    double sigma2 = TIDX(E.scal, tile, NSCAL, S_SIGMA2, lane), alpha12 = TIDX(E.scal, tile, NSCAL, S_ALPHA12, lane);
    uint32_t stayed = TIDX(E.ictr, tile, NICTR, I_STAYED, lane), bnd = TIDX(E.ictr, tile, NICTR, I_BNDSTAYED, lane);
    uint32_t chainind = TIDX(E.ictr, tile, NICTR, I_CHAININD, lane), curcount = TIDX(E.ictr, tile, NICTR, I_CURCOUNT, lane);
    uint32_t status = TIDX(E.ictr, tile, NICTR, I_STATUS, lane);

    double V[D];

    for (int it = it0; it <= it1; ++it) {
        // ---- newpar = MCMC_propose(oldpar, R)
        gen_normals(g, zs_t, d, true);
#pragma unroll
        for (int k = 0; k < D; ++k) V[k] = 0.0;
        trmv_rows<D>(Rt, zs_t, d, V);
#pragma unroll
        for (int k = 0; k < D; ++k) if (k < d) { V[k] = theta_t[(size_t)k * 64] + V[k]; cand_t[(size_t)k * 64] = V[k]; }
        // ---- bounds, prior, ss, alpha, reject
        bool inb = target_inbounds<D>(E.tgt, d, V);
        double pri2 = target_prior<D>(E.tgt, d, V);
        double ss2 = target_ss<D>(E.tgt, d, V, cand_t);
        bool reject;
        if (!inb) {
            bnd += 1; reject = true;
            if (E.method != M_RAM) alpha12 = 0.0;        // RAM leaves alpha12 stale: MCMC_run_ram.F90:52-54
        } else {
            alpha12 = d_alpha(ss1, pri1, ss2, pri2, sigma2);
            reject = true;                                // MCMC_reject, MCMC_DRAM.F90:140-155
            if (alpha12 >= 1.0) reject = false;
            else if (alpha12 > 0.0) { double u = rng_uniform(g); if (u <= alpha12) reject = false; }
        }
        if (reject) { stayed += 1; curcount += 1; }
        else { ss1 = ss2; pri1 = pri2; chainind += 1; curcount = 1; }
        // ---- MCMC_updatesigma2 (MCMC_DRAM.F90:192-206)
        if (E.updatesigma) {
            double gm = rng_gamma(g, E.gam_shape, 2.0 / (E.N0S02 + ss1));
            sigma2 = 1.0 / gm;
        }
        // ---- oldpar = newpar; MCMC_savechain (MCMC_aux.F90:167-185): accept ballot + accepted row into the ring
        unsigned long long ballot = __ballot(!reject);
        const int slot = it % E.wcap;
        if (ballot != 0ull) {
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < d) V[k] = cand_t[(size_t)k * 64];
            if (!reject) {
#pragma unroll
                for (int k = 0; k < D; ++k) if (k < d) theta_w[(size_t)k * 64] = V[k];
                if (E.hist) {
                    double *h = E.hist + ((size_t)tile * E.wcap + slot) * (size_t)(d + 1) * 64 + lane;
#pragma unroll
                    for (int k = 0; k < D; ++k) if (k < d) h[(size_t)k * 64] = V[k];
                    h[(size_t)d * 64] = ss1;
                }
            }
        }
        if (E.hist) {
            if (lane == 0) E.wacc[(size_t)tile * E.wcap + slot] = ballot;
            if (E.record_s2) E.s2hist[((size_t)tile * E.wcap + slot) * 64 + lane] = sigma2;
        }
        if (E.accmask && lane == 0) E.accmask[(size_t)(it - 1) * E.ntiles + tile] = ballot;
        // ---- MCMC_adapt_ram
        if (E.method == M_RAM && E.doadapt != 0 && !(it < E.burnintime && E.doburnin != 0)) {
            double a = ramscale[it - it0] * (alpha12 - E.alphatarget);
            ram_update<D>(Rt, zs_t, cs_t, d, a, true, V, status);
        }
    }

    TIDX(E.rngn, tile, 1, 0, lane) = g.n;
    TIDX(E.ictr, tile, NICTR, I_SAVED, lane) = (uint32_t)g.saved;
    TIDX(E.scal, tile, NSCAL, S_SAVEDY, lane) = g.saved_y;
    TIDX(E.scal, tile, NSCAL, S_SS1, lane) = ss1; TIDX(E.scal, tile, NSCAL, S_PRI1, lane) = pri1;
    TIDX(E.scal, tile, NSCAL, S_SIGMA2, lane) = sigma2; TIDX(E.scal, tile, NSCAL, S_ALPHA12, lane) = alpha12;
    TIDX(E.ictr, tile, NICTR, I_STAYED, lane) = stayed; TIDX(E.ictr, tile, NICTR, I_BNDSTAYED, lane) = bnd;
    TIDX(E.ictr, tile, NICTR, I_CHAININD, lane) = chainind; TIDX(E.ictr, tile, NICTR, I_CURCOUNT, lane) = curcount;
    TIDX(E.ictr, tile, NICTR, I_STATUS, lane) = status;
}

// ---------------------------------------------------------------- first point (MCMC_run.F90:33-39)
template <int D>
__global__ __launch_bounds__(64) void init_kernel(EngineDev E)
{
    const int lane = threadIdx.x, tile = blockIdx.x, d = E.d;
    double *theta_t = E.theta + (size_t)tile * d * 64 + lane;
    double *cand_t = E.cand + (size_t)tile * d * 64 + lane;
    double V[D];
#pragma unroll
    for (int k = 0; k < D; ++k) if (k < d) { V[k] = theta_t[(size_t)k * 64]; cand_t[(size_t)k * 64] = V[k]; }
    double pri1 = target_prior<D>(E.tgt, d, V);
    double ss1 = target_ss<D>(E.tgt, d, V, cand_t);
    TIDX(E.scal, tile, NSCAL, S_SS1, lane) = ss1; TIDX(E.scal, tile, NSCAL, S_PRI1, lane) = pri1;
    // row 1 of the chain: iteration 1 counts as accepted
    const int slot = 1 % E.wcap;
    if (E.hist) {
        double *h = E.hist + ((size_t)tile * E.wcap + slot) * (size_t)(d + 1) * 64 + lane;
        for (int k = 0; k < d; ++k) h[(size_t)k * 64] = theta_t[(size_t)k * 64];
        h[(size_t)d * 64] = ss1;
        if (lane == 0) E.wacc[(size_t)tile * E.wcap + slot] = ~0ull;
        if (E.record_s2) E.s2hist[((size_t)tile * E.wcap + slot) * 64 + lane] = TIDX(E.scal, tile, NSCAL, S_SIGMA2, lane);
    }
    if (E.accmask && lane == 0) E.accmask[tile] = ~0ull;
    for (int k = 0; k < d; ++k) TIDX(E.basetheta, tile, d, k, lane) = theta_t[(size_t)k * 64];
}

// ---------------------------------------------------------------- MCMC_adapt (MCMC_adapt.F90:12-174) at a tick
// mode bits chosen by the host from (simuind, namelist): see mcmcx_run.
enum { AD_BURN = 1, AD_AM = 2, AD_FIRST = 4 };

// dpotf2('U') on the packed matrix in Ct (holds C on entry, the factor on exit), then commit
// R = R0*2.4/sqrt(d) (MCMC_calculate_R, MCMC_adapt.F90:181-230, Cholesky path).  Returns info.
template <int D>
MCX_DEV int calculate_R(double *Ct, double *Rt, int d, int P, bool act, double (&A)[D])
{
    int info = 0;
    for (int j = 0; j < d; ++j) {
        double dot = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) if (i < j) { A[i] = Ct[(size_t)pidx(i, j, d) * 64]; dot = dfma(A[i], A[i], dot); }
        double *rowj = Ct + (size_t)rowstart(j, d) * 64;
        double ajj = rowj[0] - dot;
        bool ok = (ajj > 0.0);
        if (act && info == 0 && !ok) { info = j + 1; }
        bool go = act && info == 0;
        double rj = sqrt(ajj);
        if (go) rowj[0] = rj;
        double rinv = 1.0 / rj;
        for (int k = j + 1; k < d; ++k) {
            if (go) {
                double t = 0.0;
#pragma unroll
                for (int i = 0; i < D; ++i) if (i < j) t = dfma(Ct[(size_t)pidx(i, k, d) * 64], A[i], t);
                rowj[(size_t)(k - j) * 64] = (rowj[(size_t)(k - j) * 64] - t) * rinv;
            }
        }
    }
    if (act && info == 0) {
        double sq = sqrt((double)d);
        for (int e = 0; e < P; ++e) Rt[(size_t)e * 64] = Ct[(size_t)e * 64] * 2.4 / sq;
    }
    return info;
}

template <int D>
__global__ __launch_bounds__(64) void adapt_kernel(EngineDev E, int it, int mode)
{
    const int lane = threadIdx.x, tile = blockIdx.x, d = E.d, P = E.P;
    double *Rt = E.R + (size_t)tile * P * 64 + lane;
    double *Ct = E.cmat + (size_t)tile * P * 64 + lane;
    double *Tt = E.Rtmp + (size_t)tile * P * 64 + lane;
    double *mean_t = E.mean + (size_t)tile * d * 64 + lane;
    double *base_t = E.basetheta + (size_t)tile * d * 64 + lane;
    double *theta_t = E.theta + (size_t)tile * d * 64 + lane;
    double *dl_t = E.cand + (size_t)tile * d * 64 + lane;            // scratch for the centred row
    uint64_t *rows = E.rowlist + (size_t)tile * (E.wcap + 1) * 64 + lane;
    uint32_t stayed = TIDX(E.ictr, tile, NICTR, I_STAYED, lane);
    uint32_t curcount = TIDX(E.ictr, tile, NICTR, I_CURCOUNT, lane);
    uint32_t lastfreq = TIDX(E.ictr, tile, NICTR, I_LASTFREQ, lane);
    uint32_t basecnt = TIDX(E.ictr, tile, NICTR, I_BASECNT, lane);
    uint32_t winstart = TIDX(E.ictr, tile, NICTR, I_WINSTART, lane);
    double wsum = TIDX(E.scal, tile, NSCAL, S_WSUM, lane);
    double A[D], B[D];
    bool docalc = false;          // lanes that go on to MCMC_calculate_R

    if (mode & AD_BURN) {                                             // MCMC_adapt.F90:60-102
        double staypc = (double)stayed / (double)it;
        double sf = E.scalefactor;
        if (staypc > 1.0 - E.scalelimit) {
            for (int e = 0; e < P; ++e) Rt[(size_t)e * 64] = Rt[(size_t)e * 64] / sf;
        } else if (staypc < E.scalelimit) {
            for (int e = 0; e < P; ++e) Rt[(size_t)e * 64] = Rt[(size_t)e * 64] * sf;
        } else {
            // lastind = chainind: the covariance window restarts at the current row (lastfreq is NOT touched)
            docalc = true;
            for (int k = 0; k < d; ++k) base_t[(size_t)k * 64] = theta_t[(size_t)k * 64];
            basecnt = curcount; winstart = (uint32_t)(it + 1);
        }
    } else if (mode & AD_AM) {                                        // MCMC_adapt.F90:105-159, adapthist <= 1
        docalc = true;
        if (mode & AD_FIRST) {
            for (int e = 0; e < P; ++e) Ct[(size_t)e * 64] = E.cmat0p[e];
            for (int k = 0; k < d; ++k) mean_t[(size_t)k * 64] = E.par0[k];
        }
        // ---- phase 1: rows of chain(lastind:chainind) and their weights, from the accept ballots
        int nr = 0;
        {
            uint32_t w = basecnt;                 // count of the base row when the window started
            uint32_t slot_prev = 0xffffffffu;     // base row lives in basetheta
            for (int t = (int)winstart; t <= it; ++t) {
                const int slot = t % E.wcap;
                unsigned long long m = E.wacc[(size_t)tile * E.wcap + slot];
                if ((m >> lane) & 1ull) {
                    uint32_t wr = (nr == 0) ? (w - lastfreq) : w;
                    rows[(size_t)nr * 64] = (uint64_t)slot_prev | ((uint64_t)wr << 32);
                    ++nr; slot_prev = (uint32_t)slot; w = 1;
                } else w += 1;
            }
            uint32_t wr = (nr == 0) ? (w - lastfreq) : w;
            rows[(size_t)nr * 64] = (uint64_t)slot_prev | ((uint64_t)wr << 32);
            ++nr;
        }
        int nrmax = nr;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { int other = __shfl_xor(nrmax, o); nrmax = other > nrmax ? other : nrmax; }

        if (wsum > 0.0) {
            // ---- covmat(update=.true.): weighted Welford, one row at a time (matutils.F90:283-310)
            for (int r = 0; r < nrmax; ++r) {
                bool act = r < nr;
                uint64_t e = act ? rows[(size_t)r * 64] : 0ull;
                uint32_t slot = (uint32_t)e; double w3 = (double)(uint32_t)(e >> 32);
                const double *src = (slot == 0xffffffffu) ? base_t
                    : E.hist + ((size_t)tile * E.wcap + slot) * (size_t)(d + 1) * 64 + lane;
                if (act) {
#pragma unroll
                    for (int k = 0; k < D; ++k) if (k < d) { A[k] = src[(size_t)k * 64] - mean_t[(size_t)k * 64]; dl_t[(size_t)k * 64] = A[k]; }
                    double f1 = w3 / (wsum + w3 - 1.0);
                    double f2 = wsum / (wsum + w3);
                    for (int a = 0; a < d; ++a) {             // row a of the upper triangle: elements (a, b >= a)
                        double da = dl_t[(size_t)a * 64];
                        double *rowa = Ct + (size_t)rowstart(a, d) * 64;
#pragma unroll
                        for (int b = 0; b < D; ++b) {
                            if (b >= a && b < d) {
                                double o = da * A[b];
                                double cab = rowa[(size_t)(b - a) * 64];
                                rowa[(size_t)(b - a) * 64] = cab + f1 * (f2 * o - cab);
                            }
                        }
                    }
                    double f3 = w3 / (wsum + w3);
#pragma unroll
                    for (int k = 0; k < D; ++k) if (k < d) mean_t[(size_t)k * 64] = mean_t[(size_t)k * 64] + f3 * A[k];
                    wsum = w3 + wsum;
                }
            }
        } else {
            // ---- covmat batch branch (matutils.F90:311-338): wsum == 0 on entry
            double wsum2 = 0.0;
            for (int r = 0; r < nr; ++r) wsum2 = wsum2 + (double)(uint32_t)(rows[(size_t)r * 64] >> 32);
#pragma unroll
            for (int k = 0; k < D; ++k) B[k] = 0.0;
            for (int r = 0; r < nrmax; ++r) {
                bool act = r < nr;
                uint64_t e = act ? rows[(size_t)r * 64] : 0ull;
                uint32_t slot = (uint32_t)e; double w = (double)(uint32_t)(e >> 32);
                const double *src = (slot == 0xffffffffu) ? base_t
                    : E.hist + ((size_t)tile * E.wcap + slot) * (size_t)(d + 1) * 64 + lane;
                if (act) {
#pragma unroll
                    for (int k = 0; k < D; ++k) if (k < d) B[k] = B[k] + src[(size_t)k * 64] * w;
                }
            }
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < d) B[k] = B[k] / wsum2;          // xmean2
            for (int e = 0; e < P; ++e) Ct[(size_t)e * 64] = 0.0;
            for (int r = 0; r < nrmax; ++r) {
                bool act = r < nr;
                uint64_t e = act ? rows[(size_t)r * 64] : 0ull;
                uint32_t slot = (uint32_t)e; double w = (double)(uint32_t)(e >> 32);
                const double *src = (slot == 0xffffffffu) ? base_t
                    : E.hist + ((size_t)tile * E.wcap + slot) * (size_t)(d + 1) * 64 + lane;
                if (act) {
#pragma unroll
                    for (int k = 0; k < D; ++k) if (k < d) { A[k] = src[(size_t)k * 64] - B[k]; dl_t[(size_t)k * 64] = A[k]; }
                    // reference: cmat(i,j), j <= i = sum_r (x_ri - m_i) * ((x_rj - m_j) * w_r); kept at packed (j,i)
                    for (int j = 0; j < d; ++j) {
                        double xb = dl_t[(size_t)j * 64] * w;
                        double *rowj = Ct + (size_t)rowstart(j, d) * 64;
#pragma unroll
                        for (int i = 0; i < D; ++i) {
                            if (i >= j && i < d) rowj[(size_t)(i - j) * 64] = rowj[(size_t)(i - j) * 64] + A[i] * xb;
                        }
                    }
                }
            }
            for (int e = 0; e < P; ++e) Ct[(size_t)e * 64] = Ct[(size_t)e * 64] / (wsum2 - 1.0);
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < d) mean_t[(size_t)k * 64] = B[k];
            wsum = wsum2;
        }
        // lastfreq = count of the current row; lastind = chainind -> window restarts here
        lastfreq = curcount;
        for (int k = 0; k < d; ++k) base_t[(size_t)k * 64] = theta_t[(size_t)k * 64];
        basecnt = curcount; winstart = (uint32_t)(it + 1);
    }

    if (__any(docalc)) {
        if (docalc) for (int e = 0; e < P; ++e) Tt[(size_t)e * 64] = Ct[(size_t)e * 64];
        int info = calculate_R<D>(Tt, Rt, d, P, docalc, A);
        if (docalc) {
            TIDX(E.ictr, tile, NICTR, I_INFO, lane) = (uint32_t)info;
            if (info != 0) TIDX(E.ictr, tile, NICTR, I_STATUS, lane) |= ST_CHOL_FAIL;   // warning, old R kept (:168-171)
        }
    }
    TIDX(E.ictr, tile, NICTR, I_LASTFREQ, lane) = lastfreq;
    TIDX(E.ictr, tile, NICTR, I_BASECNT, lane) = basecnt;
    TIDX(E.ictr, tile, NICTR, I_WINSTART, lane) = winstart;
    TIDX(E.scal, tile, NSCAL, S_WSUM, lane) = wsum;
}

// ---------------------------------------------------------------- pooled moments of the current states
// out[tile][1 + d + d(d+1)/2]: partial sums over the 64 lanes of a tile by an xor-butterfly (a fixed
// pairwise tree: adjacent lanes first); the host finishes the tree over tiles, RCCL over GPUs.
// Second moments are indexed j(j+1)/2 + i for i <= j.
__global__ __launch_bounds__(64) void moments_kernel(EngineDev E, double *out, int nchains)
{
    const int lane = threadIdx.x, tile = blockIdx.x, d = E.d, P = E.P;
    const double *theta_t = E.theta + (size_t)tile * d * 64 + lane;
    const bool act = (tile * 64 + lane) < nchains;
    double *o = out + (size_t)tile * (1 + d + P);
    auto wsum64 = [](double v) {
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) v = v + __shfl_xor(v, s);
        return v;
    };
    double cnt = wsum64(act ? 1.0 : 0.0);
    if (lane == 0) o[0] = cnt;
    for (int j = 0; j < d; ++j) {
        double vj = act ? (theta_t[(size_t)j * 64] - E.par0[j]) : 0.0;
        double s1 = wsum64(vj);
        if (lane == 0) o[1 + j] = s1;
        for (int i = 0; i <= j; ++i) {
            double vi = act ? (theta_t[(size_t)i * 64] - E.par0[i]) : 0.0;
            double s2 = wsum64(vi * vj);
            if (lane == 0) o[1 + d + j * (j + 1) / 2 + i] = s2;
        }
    }
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
