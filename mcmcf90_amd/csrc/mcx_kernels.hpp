// mcx_kernels.hpp -- HIP kernels of the adaptive-Metropolis engine (gfx950, wave64).
//
// Execution model: one lane = one chain, one 64-lane wave (= one workgroup) = one
// "tile" of 64 chains.  All per-chain arrays in HBM are tile-interleaved,
//     element k of chain (tile, lane)  ->  base[(tile*K + k)*64 + lane],
// i.e. parameter-major inside a tile, so every wave access is one contiguous
// 512-byte segment and a tile's whole Cholesky factor is one sequential stream.
// Per-lane d-vectors live in VGPR arrays with compile-time indices: O(d^2) loops
// are written "runtime outer index, unrolled + guarded inner index".
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
    double *theta, *cand, *scal, *R, *R2, *iC, *Rtmp;
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

MCX_DEV int pk(int i, int j) { return j * (j + 1) / 2 + i; }     // packed upper, column-major, i <= j

// ---------------------------------------------------------------- targets (user ssfunction / priorfun / checkbounds)
template <int D>
MCX_DEV double target_ss(const DevTarget &t, int d, const double (&th)[D], double (&v)[D],
                         const double *cand_t /* tile base + lane, element stride 64 */)
{
    double ss = 0.0;
    if (t.kind == TGT_GAUSS) {
        // ss = (th-mu)' Lam (th-mu): y_i = sum_j lam(i,j) v_j ascending (fma chain), ss = sum_i y_i v_i (fma chain)
#pragma unroll
        for (int j = 0; j < D; ++j) if (j < d) v[j] = th[j] - t.mu[j];
        for (int i = 0; i < d; ++i) {
            const double *__restrict__ row = t.lam + (size_t)i * d;
            double y = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                if (j < d) { if (j == 0) y = row[0] * v[0]; else y = dfma(row[j], v[j], y); }
            }
            double vi = cand_t[(size_t)i * 64] - t.mu[i];
            if (i == 0) ss = y * vi; else ss = dfma(y, vi, ss);
        }
    } else if (t.kind == TGT_BANANA) {
        double t1 = th[0] * th[0];
        double q = dfma(t.b, t1, (D > 1) ? th[D > 1 ? 1 : 0] : 0.0) - 100.0 * t.b;
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

// ---------------------------------------------------------------- normals into LDS (mcmcrand.F90:60-83,166-190)
// Each lane appends accepted polar pairs to its own column of zbuf until it has d deviates; the
// wave loops until every participating lane is done.  The cached second deviate is honoured.
MCX_DEV void gen_normals(Rng &g, double *zbuf /* LDS [D][64] */, int lane, int d, bool participate)
{
    int k = 0;
    if (participate && g.saved && d > 0) { zbuf[0 * 64 + lane] = g.saved_y; g.saved = 0; k = 1; }
    bool need = participate && (k < d);
    while (__any(need)) {
        if (need) {
            double a, b;
            if (polar_try(g, a, b)) {
                zbuf[k * 64 + lane] = a; ++k;
                if (k < d) { zbuf[k * 64 + lane] = b; ++k; }
                else { g.saved_y = b; g.saved = 1; }
            }
            need = (k < d);
        }
    }
}

// ---------------------------------------------------------------- proposal: cand = theta + R'z  (MCMC_DRAM.F90:20-31)
// dtrmv('U','T','N') in netlib order (matutils.F90:108-109): column j = d-1..0, temp = z_j R(j,j),
// then i = j-1..0 as an fma chain.  R packed upper, column j contiguous.
template <int D>
MCX_DEV void propose(const double *Rt, const double *theta_t, double *cand_t, const double (&z)[D], int d, bool act)
{
    for (int j = d - 1; j >= 0; --j) {
        const double *col = Rt + (size_t)(j * (j + 1) / 2) * 64;
        double temp = 0.0;
        if (act) {
#pragma unroll
            for (int i = D - 1; i >= 0; --i) {
                if (i == j) temp = z[i] * col[(size_t)i * 64];
                else if (i < j) temp = dfma(col[(size_t)i * 64], z[i], temp);
            }
            cand_t[(size_t)j * 64] = theta_t[(size_t)j * 64] + temp;
        }
    }
}

// ---------------------------------------------------------------- RAM rank-1 adaptation (MCMC_run_ram.F90:104-179)
// a >= 0: cholupdate = DCHUD (dchud.f:122-139); a < 0: choldowndate = DCHDD (dchdd.f:141-179).
// c[] and s[] are the rotation vectors; x_j = z_j / sum(z^2) * |a| is formed on the fly from zbuf.
template <int D>
MCX_DEV void ram_update(double *Rt, const double *zbuf, int lane, int d, double a, bool act,
                        double (&c)[D], double (&s)[D], uint32_t &status)
{
    double su = 0.0;
    for (int k = 0; k < d; ++k) { double zk = zbuf[k * 64 + lane]; su = su + zk * zk; }
    bool up = act && (a >= 0.0);
    bool down = act && !(a >= 0.0);
    if (__any(up)) {
        if (up) {
            double pc = 0.0, ps = 0.0;           // rotation of the previous column, committed lazily
            for (int j = 0; j < d; ++j) {
                double *col = Rt + (size_t)(j * (j + 1) / 2) * 64;
                double xj = zbuf[j * 64 + lane] / su * a;
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    if (i < j) {
                        if (i == j - 1) { c[i] = pc; s[i] = ps; }
                        double rij = col[(size_t)i * 64];
                        double t = c[i] * rij + s[i] * xj;
                        xj = c[i] * xj - s[i] * rij;
                        col[(size_t)i * 64] = t;
                    }
                }
                double r;
                d_rotg(col[(size_t)j * 64], xj, r, pc, ps);
                col[(size_t)j * 64] = r;
            }
        }
    }
    if (__any(down)) {
        if (down) {
            // solve R' a = x  (dchdd.f:141-148), s[] holds the solution
            double pend = 0.0;
            for (int j = 0; j < d; ++j) {
                const double *col = Rt + (size_t)(j * (j + 1) / 2) * 64;
                double xj = -(zbuf[j * 64 + lane] / su * a);
                double dot = 0.0;
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    if (i < j) {
                        if (i == j - 1) s[i] = pend;
                        dot = dfma(col[(size_t)i * 64], s[i], dot);
                    }
                }
                double sj = xj - dot;
                pend = sj / col[(size_t)j * 64];
            }
#pragma unroll
            for (int i = 0; i < D; ++i) if (i == d - 1) s[i] = pend;
            // norm = dnrm2(p, s) classic scale/ssq form (dchdd.f:149)
            double norm;
            if (d == 1) norm = fabs(s[0]);
            else {
                double scale = 0.0, ssq = 1.0;
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    if (i < d && s[i] != 0.0) {
                        double ax = fabs(s[i]);
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
                for (int i = D - 1; i >= 0; --i) {
                    if (i < d) {
                        double scale = alpha + fabs(s[i]);
                        double aa = alpha / scale, bb = s[i] / scale;
                        double nn = sqrt(aa * aa + bb * bb);
                        c[i] = aa / nn; s[i] = bb / nn;
                        alpha = scale * nn;
                    }
                }
                for (int j = 0; j < d; ++j) {
                    double *col = Rt + (size_t)(j * (j + 1) / 2) * 64;
                    double xx = 0.0;
#pragma unroll
                    for (int i = D - 1; i >= 0; --i) {
                        if (i <= j) {
                            double rij = col[(size_t)i * 64];
                            double t = c[i] * xx + s[i] * rij;
                            col[(size_t)i * 64] = c[i] * rij - s[i] * xx;
                            xx = t;
                        }
                    }
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
    __shared__ double zbuf[D * 64];
    const int lane = threadIdx.x, tile = blockIdx.x, d = E.d;
    const double *theta_t = E.theta + (size_t)tile * d * 64 + lane;
    double *theta_w = E.theta + (size_t)tile * d * 64 + lane;
    double *cand_t = E.cand + (size_t)tile * d * 64 + lane;
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

    double A[D], B[D];

    for (int it = it0; it <= it1; ++it) {
        // ---- newpar = MCMC_propose(oldpar, R)
        gen_normals(g, zbuf, lane, d, true);
#pragma unroll
        for (int k = 0; k < D; ++k) if (k < d) A[k] = zbuf[k * 64 + lane];
        propose<D>(Rt, theta_t, cand_t, A, d, true);
#pragma unroll
        for (int k = 0; k < D; ++k) if (k < d) B[k] = cand_t[(size_t)k * 64];
        // ---- bounds, prior, ss, alpha, reject
        bool inb = target_inbounds<D>(E.tgt, d, B);
        double pri2 = target_prior<D>(E.tgt, d, B);
        double ss2 = target_ss<D>(E.tgt, d, B, A, cand_t);
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
        else {
            ss1 = ss2; pri1 = pri2;
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < d) theta_w[(size_t)k * 64] = B[k];
            chainind += 1; curcount = 1;
        }
        // ---- MCMC_updatesigma2 (MCMC_DRAM.F90:192-206)
        if (E.updatesigma) {
            double gm = rng_gamma(g, E.gam_shape, 2.0 / (E.N0S02 + ss1));
            sigma2 = 1.0 / gm;
        }
        // ---- MCMC_savechain (MCMC_aux.F90:167-185): accept ballot + accepted row into the history ring
        unsigned long long ballot = __ballot(!reject);
        const int slot = it % E.wcap;
        if (E.hist) {
            if (!reject) {
                double *h = E.hist + ((size_t)tile * E.wcap + slot) * (size_t)(d + 1) * 64 + lane;
#pragma unroll
                for (int k = 0; k < D; ++k) if (k < d) h[(size_t)k * 64] = B[k];
                h[(size_t)d * 64] = ss1;
            }
            if (lane == 0) E.wacc[(size_t)tile * E.wcap + slot] = ballot;
            if (E.record_s2) E.s2hist[((size_t)tile * E.wcap + slot) * 64 + lane] = sigma2;
        }
        if (E.accmask && lane == 0) E.accmask[(size_t)(it - 1) * E.ntiles + tile] = ballot;
        // ---- MCMC_adapt_ram
        if (E.method == M_RAM && E.doadapt != 0 && !(it < E.burnintime && E.doburnin != 0)) {
            double a = ramscale[it - it0] * (alpha12 - E.alphatarget);
            ram_update<D>(Rt, zbuf, lane, d, a, true, A, B, status);
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
    double A[D], B[D];
#pragma unroll
    for (int k = 0; k < D; ++k) if (k < d) { B[k] = theta_t[(size_t)k * 64]; cand_t[(size_t)k * 64] = B[k]; }
    double pri1 = target_prior<D>(E.tgt, d, B);
    double ss1 = target_ss<D>(E.tgt, d, B, A, cand_t);
    TIDX(E.scal, tile, NSCAL, S_SS1, lane) = ss1; TIDX(E.scal, tile, NSCAL, S_PRI1, lane) = pri1;
    // row 1 of the chain: iteration 1 counts as accepted
    const int slot = 1 % E.wcap;
    if (E.hist) {
        double *h = E.hist + ((size_t)tile * E.wcap + slot) * (size_t)(d + 1) * 64 + lane;
#pragma unroll
        for (int k = 0; k < D; ++k) if (k < d) h[(size_t)k * 64] = B[k];
        h[(size_t)d * 64] = ss1;
        if (lane == 0) E.wacc[(size_t)tile * E.wcap + slot] = ~0ull;
        if (E.record_s2) E.s2hist[((size_t)tile * E.wcap + slot) * 64 + lane] = TIDX(E.scal, tile, NSCAL, S_SIGMA2, lane);
    }
    if (E.accmask && lane == 0) E.accmask[tile] = ~0ull;
#pragma unroll
    for (int k = 0; k < D; ++k) if (k < d) TIDX(E.basetheta, tile, d, k, lane) = B[k];
}

// ---------------------------------------------------------------- MCMC_adapt (MCMC_adapt.F90:12-174) at a tick
// mode bits chosen by the host from (simuind, namelist): see Engine::run.
enum { AD_BURN = 1, AD_AM = 2, AD_FIRST = 4 };

// dpotf2('U') on the packed matrix in Rtmp (holds C on entry), then commit R = R0*2.4/sqrt(d)
// (MCMC_calculate_R, MCMC_adapt.F90:181-230, Cholesky path).  Returns info.
template <int D>
MCX_DEV int calculate_R(double *Ct /* Rtmp tile base + lane */, double *Rt, int d, int P, bool act, double (&A)[D])
{
    int info = 0;
    for (int j = 0; j < d; ++j) {
        double *colj = Ct + (size_t)(j * (j + 1) / 2) * 64;
        double dot = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) if (i < j) { A[i] = colj[(size_t)i * 64]; dot = dfma(A[i], A[i], dot); }
        double ajj = colj[(size_t)j * 64] - dot;
        bool ok = (ajj > 0.0);
        if (act && info == 0 && !ok) { info = j + 1; }
        bool go = act && info == 0;
        double rj = sqrt(ajj);
        if (go) colj[(size_t)j * 64] = rj;
        double rinv = 1.0 / rj;
        for (int k = j + 1; k < d; ++k) {
            double *colk = Ct + (size_t)(k * (k + 1) / 2) * 64;
            if (go) {
                double t = 0.0;
#pragma unroll
                for (int i = 0; i < D; ++i) if (i < j) t = dfma(colk[(size_t)i * 64], A[i], t);
                colk[(size_t)j * 64] = (colk[(size_t)j * 64] - t) * rinv;
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
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < d) base_t[(size_t)k * 64] = theta_t[(size_t)k * 64];
            basecnt = curcount; winstart = (uint32_t)(it + 1);
        }
    } else if (mode & AD_AM) {                                        // MCMC_adapt.F90:105-159, adapthist <= 1
        docalc = true;
        if (mode & AD_FIRST) {
            wsum = TIDX(E.scal, tile, NSCAL, S_WSUM, lane);           // = initcmatn (set at init, untouched so far)
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
        // maximum row count over the wave
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
                    for (int b = 0; b < d; ++b) {
                        double db = dl_t[(size_t)b * 64];
                        double *colb = Ct + (size_t)(b * (b + 1) / 2) * 64;
#pragma unroll
                        for (int a = 0; a < D; ++a) {
                            if (a <= b) {
                                double o = A[a] * db;
                                double cab = colb[(size_t)a * 64];
                                colb[(size_t)a * 64] = cab + f1 * (f2 * o - cab);
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
                    // cmat(i,j), j <= i: sum_r (x_ri - m_i) * ((x_rj - m_j) * w_r); stored at packed (j,i)
                    for (int i = 0; i < d; ++i) {
                        double xa = dl_t[(size_t)i * 64];
                        double *coli = Ct + (size_t)(i * (i + 1) / 2) * 64;
#pragma unroll
                        for (int j = 0; j < D; ++j) {
                            if (j <= i) { double xb = A[j] * w; coli[(size_t)j * 64] = coli[(size_t)j * 64] + xa * xb; }
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
#pragma unroll
        for (int k = 0; k < D; ++k) if (k < d) base_t[(size_t)k * 64] = theta_t[(size_t)k * 64];
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
// out[tile][1 + d + P] partial sums over the 64 lanes of a tile by an xor-butterfly (fixed pairwise tree);
// the host (or an RCCL all-reduce across GPUs) finishes the sum.
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
            if (lane == 0) o[1 + d + pk(i, j)] = s2;
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
