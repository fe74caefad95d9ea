// mcx_step.hpp -- the lane-per-chain sampling kernels: MCMC_run / MCMC_run_ram / MCMC_run_er iterations (step_body), MCMC_adapt_ram with
// DCHUD / DCHDD in two sweeps over column panels (ram_update), delayed rejection (dr_body), and their __global__ entry points
// (one of the family headers mcx_kernels.hpp includes, in this order: mcx_common, mcx_products, mcx_step, mcx_scam, mcx_pooled, mcx_phase,
// mcx_adapt, mcx_svd, mcx_moments)
#pragma once
#include "mcx_products.hpp"

namespace mcx {

// ---------------------------------------------------------------- RAM rank-1 adaptation (MCMC_run_ram.F90:104-179)
// a >= 0: cholupdate = DCHUD (dchud.f:122-139); a < 0: choldowndate = DCHDD (dchdd.f:141-179), restated
// left-looking by column panels: the PW columns' work values sit in registers, the rotations of the
// rows above come back from a per-chain scratch vector cs_t = (c_0, s_0, c_1, s_1, ...).  Every element
// sees the same operations in the same order as in LINPACK's column loops.
// When `fuse` is set the sweeps also accumulate the NEXT proposal P = R_new' z_next into P_t -- update lanes with
// ascending rows, downdate lanes from the diagonal up (the two pinned dtrmv orders; pdesc says
// which one a lane's next proposal uses) -- so a wave reads and writes the factor once for its update lanes and once
// more for its downdate lanes.  Returns true for lanes whose P_t is valid.
// Rotations (c_i, s_i) of the first NLC rows are kept in LDS (lc), the rest in global scratch: row i's rotation is
// re-read by every later panel, and the early rows are the ones re-read most often.
#ifndef MCX_NLC
#define MCX_NLC 19
#endif
#ifndef MCX_MIXED_UNROLL
#define MCX_MIXED_UNROLL 1
#endif
#ifndef MCX_RAM_WAVES
#define MCX_RAM_WAVES 2
#endif
constexpr int NLC = MCX_NLC;     // 19 rows x 2 doubles x 64 lanes = 19 456 B per wave: 8 waves fill the CU's 160 KiB
// MIXED: the wave holds update AND downdate lanes (RAM near its target acceptance rate).  Stores that cover part of a
// 512-byte row segment are slow whichever lanes they are (tools/layout_probe2.hip: read all + write 22 % of the lanes
// takes longer than read all + write all), so in such a wave every lane stores in both sweeps -- the lanes a sweep does
// not concern store the value they loaded -- and each sweep writes whole segments.  A wave of one kind (the bench's
// default start: no downdates) takes the other instantiation, whose update sweep stores from inside its own branch.
template <bool MIXED, int RWT = RW>
MCX_DEV bool ram_update(double *Rt, const double *zc_t, const double *zn_t, double *cs_t, double *P_t,
                        const double *theta_t, int lane, int d, double a, double su, bool act, bool fuse, uint32_t &status,
                        double *lc, bool &pdesc)
{
    const bool up = act && (a >= 0.0);
    const bool down = act && !(a >= 0.0);
    const int rwe = (RWT == RW) ? RW : ram_panel_width(d, RWT);       // RW: panels of ten (the last one narrower); wide: equal panels
    // per-chain scratch pair k (rotation c_k, s_k; for a downdate lane first the substitution's a_k): rows < NLC in LDS
#define CS_(k, w) (*((lc && (k) < NLC) ? &lc[(2 * (k) + (w)) * 64 + lane] : &cs_t[(size_t)(2 * (k) + (w)) * 64 + lane]))
    // ---- pass A, rows ascending, one read of the factor for both kinds of lanes: update lanes rotate (DCHUD), write and
    // accumulate the next proposal; downdate lanes run the forward substitution R'a = x of DCHDD (dchdd.f:141-148, x =
    // -u/sum(u**2)*a), whose solution goes to cs_t[2i+1].  xa = DCHUD's work vector x, or the substitution's partial sums.
    if (__any(act)) {
        if (act) {
            for (int J0 = 0; J0 < d; J0 += rwe) {
                const int nw = (d - J0) < rwe ? (d - J0) : rwe;
                double xa[RWT], P[RWT];
#pragma unroll
                // x = u/sum(u**2)*a
                for (int u = 0; u < RWT; ++u) { xa[u] = up ? GV(zc_t, J0 + (u < nw ? u : nw - 1)) / su * a : 0.0; P[u] = 0.0; }
if (MIXED) {
                    // next row's loads before this row's stores (see sweep B)
                    // (c, s) of an update lane or the substitution's a_i of a downdate lane, and z_next: one row ahead as well
                    double rn[RWT], cn = 0.0, sn_ = 0.0, zn_ = 0.0;
                    if (J0 > 0) {
                        const double *sg = Rt + (size_t)J0 * 64;
#pragma unroll
                        for (int u = 0; u < RWT; ++u) rn[u] = LDNT(sg, u < nw ? u : nw - 1);
                        sn_ = CS_(0, 1);
                        if (up) { cn = CS_(0, 0); zn_ = fuse ? GV(zn_t, 0) : 0.0; }
                    }
#pragma unroll MCX_MIXED_UNROLL
                    for (int i = 0; i < J0; ++i) {                       // rows above the diagonal block
                        double *seg = Rt + (size_t)(rowstart(i, d) + J0 - i) * 64;
                        double r[RWT];
#pragma unroll
                        for (int u = 0; u < RWT; ++u) r[u] = rn[u];
                        const double c = cn, sn = sn_, zi = zn_;
                        if (i + 1 < J0) {
                            const double *sg = Rt + (size_t)(rowstart(i + 1, d) + J0 - (i + 1)) * 64;
#pragma unroll
                            for (int u = 0; u < RWT; ++u) rn[u] = LDNT(sg, u < nw ? u : nw - 1);
                            sn_ = CS_(i + 1, 1);
                            if (up) { cn = CS_(i + 1, 0); zn_ = fuse ? GV(zn_t, i + 1) : 0.0; }
                        }
                        if (up) {
#pragma unroll
                            for (int u = 0; u < RWT; ++u) {
                                double t = c * r[u] + sn * xa[u];
                                xa[u] = c * xa[u] - sn * r[u];
                                r[u] = t;
                                P[u] = dfma(t, zi, P[u]);
                            }
                        } else {
#pragma unroll
                            for (int u = 0; u < RWT; ++u) xa[u] = dfma(r[u], sn, xa[u]);
                        }
#pragma unroll
                        for (int u = 0; u < RWT; ++u) if (u < nw) STNT(seg, u, r[u]);
                    }
                } else {
#pragma unroll 2
                    for (int i = 0; i < J0; ++i) {                       // rows above the diagonal block
                        double *seg = Rt + (size_t)(rowstart(i, d) + J0 - i) * 64;
                        double r[RWT];
#pragma unroll
                        for (int u = 0; u < RWT; ++u) r[u] = LDNT(seg, u < nw ? u : nw - 1);
                        if (up) {
                            const bool inl = lc && i < NLC;
                            const double c = inl ? lc[(2 * i) * 64 + lane] : GV(cs_t, 2 * i);
                            const double sn = inl ? lc[(2 * i + 1) * 64 + lane] : GV(cs_t, 2 * i + 1);
                            const double zi = fuse ? GV(zn_t, i) : 0.0;
#pragma unroll
                            for (int u = 0; u < RWT; ++u) {
                                double t = c * r[u] + sn * xa[u];
                                xa[u] = c * xa[u] - sn * r[u];
                                if (u < nw) STNT(seg, u, t);
                                P[u] = dfma(t, zi, P[u]);
                            }
                        } else {
                            const double si = CS_(i, 1);
#pragma unroll
                            for (int u = 0; u < RWT; ++u) xa[u] = dfma(r[u], si, xa[u]);
                        }
                    }
                }
                for (int i = J0; i < J0 + nw; ++i) {                     // diagonal block
                    double *seg = Rt + (size_t)rowstart(i, d) * 64;
                    const int ui = i - J0, m = d - 1 - i;
                    double r[RWT];
#pragma unroll
                    for (int u = 0; u < RWT; ++u) { int k = u - ui; k = k < 0 ? 0 : k; k = k > m ? m : k; r[u] = LDNT(seg, k); }
                    double xi = xa[0];
#pragma unroll
                    for (int u = 1; u < RWT; ++u) xi = (u == ui) ? xa[u] : xi;
                    if (up) {
                        const double zi = fuse ? GV(zn_t, i) : 0.0;
                        double rr, c, sn;
                        d_rotg(GV(seg, 0), xi, rr, c, sn);
                        GV(seg, 0) = rr;
                        if (lc && i < NLC) { lc[(2 * i) * 64 + lane] = c; lc[(2 * i + 1) * 64 + lane] = sn; }
                        // (the last panel's rotations have no later panel to serve)
                        else if (J0 + nw < d) { GV(cs_t, 2 * i) = c; GV(cs_t, 2 * i + 1) = sn; }
#pragma unroll
                        for (int u = 0; u < RWT; ++u) {
                            const bool off = (u > ui) && (u < nw);
                            double t = c * r[u] + sn * xa[u];
                            double nx = c * xa[u] - sn * r[u];
                            xa[u] = off ? nx : xa[u];
                            if (!MIXED) { if (off) STNT(seg, u - ui, t); }
                            else r[u] = off ? t : r[u];
                            double tp = (u == ui) ? rr : t;
                            double np = dfma(tp, zi, P[u]);
                            P[u] = (u >= ui && u < nw) ? np : P[u];
                        }
                    } else {
                        double si = -(GV(zc_t, i) / su * a) - xi;
                        si = si / GV(seg, 0);
                        CS_(i, 1) = si;
#pragma unroll
                        for (int u = 0; u < RWT; ++u) { double na = dfma(r[u], si, xa[u]); xa[u] = (u > ui) ? na : xa[u]; }
                    }
                    if (MIXED) {                                         // the off-diagonal part of the row, every lane
#pragma unroll
                        for (int u = 0; u < RWT; ++u) if ((u > ui) && (u < nw)) STNT(seg, u - ui, r[u]);
                    }
                }
                if (up && fuse) {                        // next candidate = theta + R_new' z_next (MCMC_DRAM.F90:29)
                    double th[RWT];                       // the state's loads before the candidate's stores (see copy_vec)
#pragma unroll
                    for (int u = 0; u < RWT; ++u) th[u] = GV(theta_t, J0 + (u < nw ? u : nw - 1));
#pragma unroll
                    for (int u = 0; u < RWT; ++u) if (u < nw) GV(P_t, J0 + u) = th[u] + P[u];
                }
            }
        }
    }
    if (up) pdesc = false;
    bool down_ok = false;
    if (__any(down)) {
        if (down) {
            // norm = dnrm2(p, s), classic scale/ssq form (dchdd.f:149)
            double norm;
            if (d == 1) norm = fabs(CS_(0, 1));
            else {
                double scale = 0.0, ssq = 1.0;
#pragma unroll 4
                for (int k = 0; k < d; ++k) {
                    double xk = CS_(k, 1);
                    if (xk != 0.0) {
                        double ax = fabs(xk);
                        if (scale < ax) { double q = scale / ax; ssq = 1.0 + ssq * (q * q); scale = ax; }
                        else { double q = ax / scale; ssq = ssq + q * q; }
                    }
                }
                norm = scale * sqrt(ssq);
            }
            if (!(norm < 1.0)) {
                status |= ST_RAM_DOWNDATE_FAIL;      // INFO = -1: R untouched (the reference stops here)
                pdesc = false;
            } else {
                down_ok = true;
                pdesc = true;
                double alpha = sqrt(1.0 - norm * norm);
#pragma unroll 2
                for (int k = d - 1; k >= 0; --k) {   // dchdd.f:158-167
                    double sk = CS_(k, 1);
                    double scale = alpha + fabs(sk);
                    double aa = alpha / scale, bb = sk / scale;
                    double nn = sqrt(aa * aa + bb * bb);
                    CS_(k, 0) = aa / nn;
                    CS_(k, 1) = bb / nn;
                    alpha = scale * nn;
                }
            }
        }
        // ---- pass B (dchdd.f:171-179): each column from its diagonal up; the next proposal accumulates in that
        // same order (mcxo_trmv_ut_desc), so downdate lanes, too, read and write the factor once more and are done.
        // MIXED: every lane of the wave loads and stores (whole segments); only the downdate lanes change the values.
        const bool touch = MIXED ? act : down_ok;
        if (__any(down_ok)) {
            if (touch) {
                for (int J0 = 0; J0 < d; J0 += rwe) {
                    const int nw = (d - J0) < rwe ? (d - J0) : rwe;
                    double xx[RWT], P[RWT];
#pragma unroll
                    for (int u = 0; u < RWT; ++u) { xx[u] = 0.0; P[u] = 0.0; }
                    for (int i = J0 + nw - 1; i >= J0; --i) {            // diagonal block, rows descending
                        double *seg = Rt + (size_t)rowstart(i, d) * 64;
                        const int ui = i - J0, m = d - 1 - i;
                        double r[RWT];
#pragma unroll
                        for (int u = 0; u < RWT; ++u) { int k = u - ui; k = k < 0 ? 0 : k; k = k > m ? m : k; r[u] = LDB(seg, k); }
                        if (!MIXED || down_ok) {
                            const double ci = CS_(i, 0), si = CS_(i, 1);
                            const double zi = fuse ? GV(zn_t, i) : 0.0;
#pragma unroll
                            for (int u = 0; u < RWT; ++u) {
                                const bool on = (u >= ui) && (u < nw);
                                double t = ci * xx[u] + si * r[u];
                                double nr = ci * r[u] - si * xx[u];
                                if (!MIXED) { if (on) STB(seg, u - ui, nr); }
                                else r[u] = on ? nr : r[u];
                                xx[u] = on ? t : xx[u];
                                const double np = (u == ui) ? nr * zi : dfma(nr, zi, P[u]);
                                P[u] = on ? np : P[u];
                            }
                        }
                        if (MIXED) {
#pragma unroll
                            for (int u = 0; u < RWT; ++u) if ((u >= ui) && (u < nw)) STB(seg, u - ui, r[u]);
                        }
                    }
if (MIXED) {
                        // The next row's loads go out before this row's stores: vmcnt retires in order, so a load issued
                        // after a store cannot be waited for without waiting for that store's acknowledgement -- which
                        // would put the store latency on every row's critical path.
                        double rn[RWT], cn = 0.0, sn_ = 0.0, zn_ = 0.0;
                        if (J0 > 0) {
                            const double *sg = Rt + (size_t)(rowstart(J0 - 1, d) + 1) * 64;
#pragma unroll
                            for (int u = 0; u < RWT; ++u) rn[u] = LDB(sg, u < nw ? u : nw - 1);
                            if (down_ok) { cn = CS_(J0 - 1, 0); sn_ = CS_(J0 - 1, 1); zn_ = fuse ? GV(zn_t, J0 - 1) : 0.0; }
                        }
#pragma unroll MCX_MIXED_UNROLL
                        for (int i = J0 - 1; i >= 0; --i) {              // rows above, descending
                            double *seg = Rt + (size_t)(rowstart(i, d) + J0 - i) * 64;
                            double r[RWT];
#pragma unroll
                            for (int u = 0; u < RWT; ++u) r[u] = rn[u];
                            const double ci = cn, si = sn_, zi = zn_;
                            if (i > 0) {
                                const double *sg = Rt + (size_t)(rowstart(i - 1, d) + J0 - (i - 1)) * 64;
#pragma unroll
                                for (int u = 0; u < RWT; ++u) rn[u] = LDB(sg, u < nw ? u : nw - 1);
                                if (down_ok) { cn = CS_(i - 1, 0); sn_ = CS_(i - 1, 1); zn_ = fuse ? GV(zn_t, i - 1) : 0.0; }
                            }
                            if (down_ok) {
#pragma unroll
                                for (int u = 0; u < RWT; ++u) {
                                    double t = ci * xx[u] + si * r[u];
                                    const double nr = ci * r[u] - si * xx[u];
                                    r[u] = nr;
                                    xx[u] = t;
                                    P[u] = dfma(nr, zi, P[u]);
                                }
                            }
#pragma unroll
                            for (int u = 0; u < RWT; ++u) if (u < nw) STB(seg, u, r[u]);
                        }
                    } else {
#pragma unroll 2
                        for (int i = J0 - 1; i >= 0; --i) {              // rows above, descending
                            double *seg = Rt + (size_t)(rowstart(i, d) + J0 - i) * 64;
                            const double ci = CS_(i, 0), si = CS_(i, 1);
                            const double zi = fuse ? GV(zn_t, i) : 0.0;
                            double r[RWT];
#pragma unroll
                            for (int u = 0; u < RWT; ++u) r[u] = LDB(seg, u < nw ? u : nw - 1);
#pragma unroll
                            for (int u = 0; u < RWT; ++u) {
                                double t = ci * xx[u] + si * r[u];
                                const double nr = ci * r[u] - si * xx[u];
                                if (u < nw) STB(seg, u, nr);
                                xx[u] = t;
                                P[u] = dfma(nr, zi, P[u]);
                            }
                        }
                    }
                    if (fuse && down_ok) {
                        double th[RWT];
#pragma unroll
                        for (int u = 0; u < RWT; ++u) th[u] = GV(theta_t, J0 + (u < nw ? u : nw - 1));
#pragma unroll
                        for (int u = 0; u < RWT; ++u) if (u < nw) GV(P_t, J0 + u) = th[u] + P[u];
                    }
                }
            }
        }
    }
    return (up || down_ok) && fuse;
#undef CS_
}

// The same adaptation on a FULL column-major factor (condmax > 0: R is the d x d SVD factor U sqrt(s) 2.4/sqrt(d) of
// covtor, and MCMC_adapt_ram hands it to dchud / dchdd as it is, MCMC_run_ram.F90:168-172).  LINPACK only touches
// R(i,j), i <= j; the proposal matmulx(R,u) (MCMC_run_ram.F90:96-97) goes on using the whole matrix.  Plain column
// loops, the arithmetic of ram_update element for element; not fused, not tuned (the combination is a curiosity of the
// reference, kept so that every namelist it accepts runs).
#define RF(i, j) Rf_t[((size_t)(j) * d + (i)) * 64 + lane]
MCX_DEV void ram_update_full(double *Rf_t, const double *zc_t, double *cs_t, int lane, int d, double a, double su, bool act,
                             uint32_t &status)
{
    if (!act) return;
    if (a >= 0.0) {                                              // dchud.f:122-139
        for (int j = 0; j < d; ++j) {
            double xj = GV(zc_t, j) / su * a;
            for (int i = 0; i < j; ++i) {
                const double c = GV(cs_t, 2 * i), sn = GV(cs_t, 2 * i + 1), r = RF(i, j);
                double t = c * r + sn * xj;
                xj = c * xj - sn * r;
                RF(i, j) = t;
            }
            double rr, c, sn;
            d_rotg(RF(j, j), xj, rr, c, sn);
            RF(j, j) = rr; GV(cs_t, 2 * j) = c; GV(cs_t, 2 * j + 1) = sn;
        }
        return;
    }
    for (int j = 0; j < d; ++j) {                                // dchdd.f:141-148: R'a = x, x = -u/sum(u**2)*a
        double acc = 0.0;
        for (int i = 0; i < j; ++i) acc = dfma(RF(i, j), GV(cs_t, 2 * i + 1), acc);
        double xj = -(GV(zc_t, j) / su * a);
        double sj = xj - acc;
        GV(cs_t, 2 * j + 1) = sj / RF(j, j);
    }
    double norm;                                                 // dnrm2, dchdd.f:149
    if (d == 1) norm = fabs(GV(cs_t, 1));
    else {
        double scale = 0.0, ssq = 1.0;
        for (int k = 0; k < d; ++k) {
            double xk = GV(cs_t, 2 * k + 1);
            if (xk != 0.0) {
                double ax = fabs(xk);
                if (scale < ax) { double q = scale / ax; ssq = 1.0 + ssq * (q * q); scale = ax; }
                else { double q = ax / scale; ssq = ssq + q * q; }
            }
        }
        norm = scale * sqrt(ssq);
    }
    if (!(norm < 1.0)) { status |= ST_RAM_DOWNDATE_FAIL; return; }
    double alpha = sqrt(1.0 - norm * norm);
    for (int k = d - 1; k >= 0; --k) {                           // dchdd.f:158-167
        double sk = GV(cs_t, 2 * k + 1);
        double scale = alpha + fabs(sk);
        double aa = alpha / scale, bb = sk / scale;
        double nn = sqrt(aa * aa + bb * bb);
        GV(cs_t, 2 * k) = aa / nn;
        GV(cs_t, 2 * k + 1) = bb / nn;
        alpha = scale * nn;
    }
    for (int j = 0; j < d; ++j) {                                // dchdd.f:171-179
        double xx = 0.0;
        for (int i = j; i >= 0; --i) {
            const double ci = GV(cs_t, 2 * i), si = GV(cs_t, 2 * i + 1), r = RF(i, j);
            double t = ci * xx + si * r;
            RF(i, j) = ci * r - si * xx;
            xx = t;
        }
    }
}
#undef RF

// ---------------------------------------------------------------- delayed rejection (MCMC_run.F90:65-91)
// q = dx' iC dx with iC symmetric, upper triangle packed by rows (dsymv 'U' + sum, MCMC_DRAM.F90:180-182,
// matutils.F90:180): y_i = sum_j S(i,j) dx_j ascending j as an fma chain, q = sum_i y_i dx_i.
// One sweep over the rows: row i finishes y_i and feeds S(i,j) dx_i into y_j for j > i.
// X holds dx, Y the running y; both per-lane LDS vectors.
MCX_DEV double quadform_sym(const double *St, int lane, int d, const double *X, double *Y)
{
    double q = 0.0;
    for (int i = 0; i < d; ++i) {
        const double *rowp = St + (size_t)rowstart(i, d) * 64;
        const int n = d - i;
        const double dxi = XL(i);
        double sii = GV(rowp, 0);
        double yi = (i == 0) ? sii * dxi : dfma(sii, dxi, Y[i * 64 + lane]);
        // a batch of row elements at a time: the batch's dx and y values are read together, then the chain of y_i and the
        // independent updates of y_{i+k} -- element by element every update's LDS store stands between the next element's
        // loads and the ones before it (the compiler must assume the two vectors overlap), a round trip per element
        sweep_batches(rowp, lane, 1, n, [&](int k, const double (&sij)[CH], int m) {
            double xs[CH], ys[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) { const int kk = i + k + (u < m ? u : m - 1); xs[u] = XL(kk);
                ys[u] = (i == 0) ? 0.0 : Y[kk * 64 + lane]; }
#pragma unroll
            for (int u = 0; u < CH; ++u) if (u < m) yi = dfma(sij[u], xs[u], yi);
#pragma unroll
            for (int u = 0; u < CH; ++u) if (u < m) Y[(i + k + u) * 64 + lane] = (i == 0) ? sij[u] * dxi : dfma(sij[u], dxi, ys[u]);
        });
        q = q + yi * dxi;
    }
    return q;
}

// the same quadratic form with ONE inverse covariance for every chain (pooled mode with delayed rejection): Ss is the
// packed upper triangle by rows, wave-uniform, read through the scalar cache; per element the operations of quadform_sym
MCX_DEV double quadform_sym_shared(const double *__restrict__ Ss, int lane, int d, const double *X, double *Y)
{
    double q = 0.0;
    for (int i = 0; i < d; ++i) {
        const double *__restrict__ rowp = Ss + rowstart(i, d);
        const int n = d - i;
        const double dxi = XL(i);
        const double sii = rowp[0];
        double yi = (i == 0) ? sii * dxi : dfma(sii, dxi, Y[i * 64 + lane]);
        for (int k = 1; k < n; ++k) {
            const double sij = rowp[k];
            yi = dfma(sij, XL(i + k), yi);
            Y[(i + k) * 64 + lane] = (i == 0) ? sij * dxi : dfma(sij, dxi, Y[(i + k) * 64 + lane]);
        }
        q = q + yi * dxi;
    }
    return q;
}

// ---------------------------------------------------------------- the step kernel
// Iterations it0..it1 (absolute simuind) of MCMC_run (MCMC_run.F90:41-107) or MCMC_run_ram
// (MCMC_run_ram.F90:45-81) for one tile of 64 chains.  LDS is used only by the delayed-rejection
// quadratic forms (2*d*64 doubles when dodr, none otherwise).
template <bool RAM, bool DR, bool POOLED, bool WIDE_T = (RAM || (!DR && !POOLED)), bool FULLR = false, bool LDSV = false,
    bool LDSR = false, bool XG = false, int RWT = RW>
MCX_DEV void step_body(const EngineDev &E, int it0, int it1, const double *__restrict__ ramscale,
                       const double *__restrict__ g_mu, const double *__restrict__ g_lamT,
                       const double *__restrict__ g_sharedR, const double *__restrict__ g_sharedR2 = nullptr,
                       const double *__restrict__ g_sharediC = nullptr)
{
    extern __shared__ double Xlds[];
    const int lane = threadIdx.x, tile = blockIdx.x, d = E.d;
    // the delayed-rejection quadratic forms' two vectors: LDS, or (XG, a compile-time choice: no flat accesses) the chain's global scratch
    double *X = XG ? E.xscr + (size_t)tile * 2 * d * 64 : Xlds;
    // step_kernel_ldsv: launched with 4 d x 512 bytes of LDS (a compile-time choice, so that the vectors' accesses are ds_read / ds_write,
    // not flat)
    constexpr bool ldsv = LDSV && !RAM && !DR && !POOLED;
    // step_kernel_ldsr (npar <= TW): besides the state, the chain's packed factor stays in LDS for the launch -- AM only reads
    // it between two ticks -- and ONE vector serves as normals, proposal and candidate (a single column panel: the product
    // has read every normal before it stores anything)
    constexpr bool ldsr = ldsv && LDSR;
    double *theta_g = E.theta + (size_t)tile * d * 64;
    double *theta_t = ldsv ? X : theta_g;
    double *cand_t = ldsv ? X + (size_t)d * 64 : E.cand + (size_t)tile * d * 64;           // proposal vector P, then candidate theta + P
    // two normal vectors: this iteration's and the next one's
    double *zs_t = ldsr ? cand_t : (ldsv ? X + (size_t)2 * d * 64 : E.zs + (size_t)tile * 2 * d * 64);
    double *cs_t = E.cs + (size_t)tile * 2 * d * 64;           // RAM: rotations; DR: second-stage candidate
    if (ldsv) for (int k = 0; k < d; ++k) GV(theta_t, k) = GV(theta_g, k);
    // step_kernel_ram_ldsr (npar <= RW: one column panel): the factor that DCHUD / DCHDD rewrite at every iteration stays in LDS for the
    // launch, behind the 2 npar vectors of rotations -- north_star's "Cholesky factor staged in LDS" for the rank-one update itself
    constexpr bool ramr = RAM && LDSR && !FULLR;
    double *Rt = (ldsr || ramr) ? X + (size_t)2 * d * 64 : E.R + (size_t)tile * E.P * 64;
    if (ldsr || ramr) { const double *Rg = E.R + (size_t)tile * E.P * 64; copy_vec(Rt, Rg, nullptr, lane, E.P); }
    double *Y = X + (size_t)d * 64;
    double *c2_t = cs_t;

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
    uint32_t dracc = TIDX(E.ictr, tile, NICTR, I_DRACC, lane), drtries = TIDX(E.ictr, tile, NICTR, I_DRTRIES, lane);
    uint32_t erstayed = TIDX(E.ictr, tile, NICTR, I_ERSTAYED, lane);
    // the next proposal's dtrmv order (after a downdate: diagonal first)
    bool pdesc = RAM && TIDX(E.ictr, tile, NICTR, I_PDESC, lane) != 0u;
    uint32_t downs = RAM ? TIDX(E.ictr, tile, NICTR, I_DOWNS, lane) : 0u;

    bool have_p = false;                          // lanes whose candidate is already in cand_t
    double su_c = gen_normals<RAM ? 1 : MCX_RNG_NB>(g, zs_t + (ldsr ? 0 : (size_t)(it0 & 1) * d * 64), lane, d, true), su_n = 0.0;

    for (int it = it0; it <= it1; ++it) {
        double *zc_t = ldsr ? zs_t : zs_t + (size_t)(it & 1) * d * 64;          // z of this iteration
        double *zn_t = ldsr ? zs_t : zs_t + (size_t)((it + 1) & 1) * d * 64;    // z of the next one
        // ---- newpar = MCMC_propose(oldpar, R)
        if (POOLED) { if (E.usesvd) gemvN_shared(g_sharedR, zc_t, cand_t, theta_t, lane, d); else trmv_shared(g_sharedR, zc_t, cand_t,
            theta_t, lane, d); }
        else if (E.usesvd) gemvN_panels(E.Rf + (size_t)tile * d * d * 64, zc_t, cand_t, theta_t, lane, d, true);   // matmulx(R,z)
        else if (__any(!have_p)) trmv_panels<!RAM>(Rt, zc_t, cand_t, theta_t, lane, d, !have_p, RAM && pdesc);
        // ---- bounds, prior, ss, alpha, reject
        bool inb = target_inbounds(E.tgt, d, lane, cand_t);
        double pri2 = target_prior(E.tgt, d, lane, cand_t);
        double ss2 = target_ss<WIDE_T>(E.tgt, d, lane, cand_t, g_mu, g_lamT);   // wide (candidate read once) where registers allow
        bool reject;
        if (!RAM && !DR && E.method == M_ER) {            // early rejection, MCMC_run_er.F90:60-89
            if (!inb) { bnd += 1; reject = true; }
            else {
                double u = rng_uniform(g);                // MCMC_sscrit, MCMC_DRAM.F90:124-135: always drawn
                double sscrit = -2.0 * d_log(u) + ss1 / sigma2 + pri1;
                if (pri2 >= sscrit) { reject = true; erstayed += 1; }
                else { sscrit = sigma2 * (sscrit - pri2); reject = (ss2 >= sscrit); }
            }
        } else if (!inb) {
            if (!DR) bnd += 1;                            // MCMC_run.F90:49
            reject = true;
            if (!RAM) alpha12 = 0.0;                      // RAM leaves alpha12 stale: MCMC_run_ram.F90:52-54
        } else {
            alpha12 = d_alpha(ss1, pri1, ss2, pri2, sigma2);
            reject = true;                                // MCMC_reject, MCMC_DRAM.F90:140-155
            if (alpha12 >= 1.0) reject = false;
            else if (alpha12 > 0.0) { double u = rng_uniform(g); if (u <= alpha12) reject = false; }
        }
        // ---- second stage: one delayed-rejection try with R2 = R/drscale (MCMC_run.F90:65-91)
        bool dr_moved = false;
        if (DR && __any(reject)) {
            const bool m = reject;
            if (m) drtries += 1;
            double *z2_t = zn_t;                          // stage-2 normals: the "next" buffer is still free
            gen_normals<RAM ? 1 : MCX_RNG_NB>(g, z2_t, lane, d, m);
            // one R2 for every chain; lanes that did not draw compute on stale normals and are not looked at
            if (POOLED) {
                if (E.usesvd) gemvN_shared(g_sharedR2, z2_t, c2_t, theta_t, lane, d); else trmv_shared(g_sharedR2, z2_t, c2_t, theta_t,
                    lane, d);
            }
            else if (E.usesvd) gemvN_panels(E.R2f + (size_t)tile * d * d * 64, z2_t, c2_t, theta_t, lane, d, m);
            else trmv_panels<true>(E.R2 + (size_t)tile * E.P * 64, z2_t, c2_t, theta_t, lane, d, m);
            if (m) {
                bool inb2 = target_inbounds(E.tgt, d, lane, c2_t);
                if (!inb2) bnd += 1;
                else {
                    double pri3 = target_prior(E.tgt, d, lane, c2_t);
                    double ss3 = target_ss<false>(E.tgt, d, lane, c2_t, g_mu, g_lamT);
                    // MCMC_DR_alpha13, MCMC_DRAM.F90:162-186
                    double alpha32;
                    if (alpha12 == 0.0) alpha32 = 0.0;
                    else alpha32 = min1(d_exp(-0.5 * ((ss2 - ss3) / sigma2 + (pri2 - pri3))));
                    double l2 = -0.5 * ((ss3 - ss1) / sigma2 + (pri3 - pri1));
                    const double *iCt = POOLED ? nullptr : E.iC + (size_t)tile * E.P * 64;
                    for (int k = 0; k < d; ++k) XL(k) = GV(c2_t, k) - GV(cand_t, k);
                    double qa = POOLED ? quadform_sym_shared(g_sharediC, lane, d, X, Y) : quadform_sym(iCt, lane, d, X, Y);
                    for (int k = 0; k < d; ++k) XL(k) = GV(theta_t, k) - GV(cand_t, k);
                    double qb = POOLED ? quadform_sym_shared(g_sharediC, lane, d, X, Y) : quadform_sym(iCt, lane, d, X, Y);
                    double q1 = -0.5 * (qa - qb);
                    double alpha13 = min1(d_exp(l2 + q1) * (1.0 - alpha32) / (1.0 - alpha12));
                    bool rej2 = true;
                    if (alpha13 >= 1.0) rej2 = false;
                    else if (alpha13 > 0.0) { double u = rng_uniform(g); if (u <= alpha13) rej2 = false; }
                    if (!rej2) { dracc += 1; reject = false; dr_moved = true; ss2 = ss3; pri2 = pri3; }
                }
            }
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
        if (!reject) {
            double *h = E.hist ? E.hist + ((size_t)tile * E.wcap + slot) * (size_t)E.hs * 64 : nullptr;
            const double *src = dr_moved ? c2_t : cand_t;     // newpar = newpar2 when the DR try was accepted
            copy_vec(theta_t, src, h, lane, d);
            if (h) GV(h, d) = ss1;
        }
        if (E.hist) {
            if (lane == 0) E.wacc[(size_t)tile * E.wcap + slot] = ballot;
            if (E.record_s2) E.s2hist[((size_t)tile * E.wcap + slot) * 64 + lane] = sigma2;
        }
        if (E.accmask && lane == 0) E.accmask[(size_t)(it - 1) * E.ntiles + tile] = ballot;
        // ---- the next iteration's normals: nothing else draws between here and its MCMC_propose
        const bool pre = (it < it1);
        if (pre) su_n = gen_normals<RAM ? 1 : MCX_RNG_NB>(g, zn_t, lane, d, true);
        // ---- MCMC_adapt_ram
        have_p = false;
        if (RAM && E.doadapt != 0 && !(it < E.burnintime && E.doburnin != 0)) {
            double a = ramscale[it - it0] * (alpha12 - E.alphatarget);
            downs += (a >= 0.0) ? 0u : 1u;
            if (FULLR) ram_update_full(E.Rf + (size_t)tile * d * d * 64, zc_t, cs_t, lane, d, a, su_c, true, status);   // condmax > 0
            else if (__any(!(a >= 0.0)))                  // a wave with downdate lanes: whole-segment stores in both sweeps
                have_p = ram_update<true, RWT>(Rt, zc_t, zn_t, cs_t, cand_t, theta_t, lane, d, a, su_c, true, pre, status,
                    RAM ? X : nullptr, pdesc);
            else have_p = ram_update<false, RWT>(Rt, zc_t, zn_t, cs_t, cand_t, theta_t, lane, d, a, su_c, true, pre, status,
                RAM ? X : nullptr, pdesc);
        }
        su_c = su_n;
    }

    TIDX(E.rngn, tile, 1, 0, lane) = g.n;
    TIDX(E.ictr, tile, NICTR, I_SAVED, lane) = (uint32_t)g.saved;
    TIDX(E.scal, tile, NSCAL, S_SAVEDY, lane) = g.saved_y;
    TIDX(E.scal, tile, NSCAL, S_SS1, lane) = ss1; TIDX(E.scal, tile, NSCAL, S_PRI1, lane) = pri1;
    TIDX(E.scal, tile, NSCAL, S_SIGMA2, lane) = sigma2; TIDX(E.scal, tile, NSCAL, S_ALPHA12, lane) = alpha12;
    TIDX(E.ictr, tile, NICTR, I_STAYED, lane) = stayed; TIDX(E.ictr, tile, NICTR, I_BNDSTAYED, lane) = bnd;
    TIDX(E.ictr, tile, NICTR, I_CHAININD, lane) = chainind; TIDX(E.ictr, tile, NICTR, I_CURCOUNT, lane) = curcount;
    TIDX(E.ictr, tile, NICTR, I_STATUS, lane) = status;
    TIDX(E.ictr, tile, NICTR, I_DRACC, lane) = dracc; TIDX(E.ictr, tile, NICTR, I_DRTRIES, lane) = drtries;
    TIDX(E.ictr, tile, NICTR, I_ERSTAYED, lane) = erstayed;
    if (RAM) { TIDX(E.ictr, tile, NICTR, I_PDESC, lane) = pdesc ? 1u : 0u; TIDX(E.ictr, tile, NICTR, I_DOWNS, lane) = downs; }
    if (ldsv) for (int k = 0; k < d; ++k) GV(theta_g, k) = GV(theta_t, k);
    if (ramr) { double *Rg = E.R + (size_t)tile * E.P * 64; copy_vec(Rg, Rt, nullptr, lane, E.P); }
}

// ---------------------------------------------------------------- delayed rejection with per-chain factors
// MCMC_run with drscale > 0 (MCMC_run.F90:41-107).  step_body<DR> above moves, per iteration, the factor R, its
// second-stage copy R2, the inverse covariance iC TWICE (one sweep per quadratic form of MCMC_DR_alpha13) and ~2.5 kB of
// per-chain scratch vectors that L2 cannot hold between a write and the read that follows it (rocprof,
// profiles/r02_d_final/c3_dram: 9.4 kB per iteration against 3.7 kB of factors at npar = 20).  Here
//   * the normals and the first-stage candidate live in the wave's two LDS vectors (the ones step_body<DR> keeps for the
//     quadratic forms); only the second-stage candidate goes through the chain's global scratch;
//   * the two quadratic forms dx' iC dx of MCMC_DR_alpha13 (MCMC_DRAM.F90:180-182) share ONE sweep over iC
//     (quadform2_panels): both dx vectors in the two LDS vectors (normals and first-stage candidate are dead by then).
// Every chain of operations is the one of step_body<DR> (same operands, same order): the results are its bit for bit.

// qa = xa' S xa and qb = xb' S xb, S symmetric with its upper triangle packed by rows, in one sweep over S.
// Per form the operations of quadform_sym: y_i = sum_j S(i,j) x_j as ONE fma chain ascending in j -- first the column part
// S(i',i) x_i' (i' < i), then the diagonal, then the row part -- and q = sum_i y_i x_i ascending in i.  Column panels of TQ:
// the column parts of the panel's y_j accumulate in registers while the rows stream by (rows ascending); a row's own
// chain y_i runs along the row, across the panels, and waits between two panels in the chain's global scratch (ysa, ysb:
// npar doubles each, one store and one load per row and panel boundary -- nothing at npar <= TQ); q takes y_i x_i when
// the last panel completes it, rows ascending.  xa, xb: per-lane LDS vectors.
constexpr int TQ = 10;
MCX_DEV void quadform2_panels(const double *St, int lane, int d, const double *Xa, const double *Xb, double *ysa, double *ysb,
                              double &qa, double &qb)
{
    qa = 0.0; qb = 0.0;
    for (int J0 = 0; J0 < d; J0 += TQ) {
        const int nw = (d - J0) < TQ ? (d - J0) : TQ;
        const bool last = J0 + TQ >= d;
        double Ya[TQ], Yb[TQ], xja[TQ], xjb[TQ];
#pragma unroll
        for (int u = 0; u < TQ; ++u) { const int j = J0 + (u < nw ? u : nw - 1); xja[u] = GV(Xa, j); xjb[u] = GV(Xb, j); Ya[u] = 0.0;
            Yb[u] = 0.0; }
        // rows above the panel: the row's chain takes the panel's nw elements, the panel's columns take the row's x_i
        {
#ifndef MCX_Q2_NB
#define MCX_Q2_NB 1
#endif
            // rows in flight: more than one spills registers (2: 70, 3: 167), and a spill here costs more than the latency it hides (c3:
            // 20.3 / 17.5 / 14.8 ms per launch at 3 / 2 / 1)
            constexpr int NB = MCX_Q2_NB;
            double rr[NB][TQ], xa_[NB], xb_[NB], ya_[NB], yb_[NB];
#define MCX_Q2_LD(s_, i_) { const double *seg_ = St + (size_t)(rowstart((i_), d) + J0 - (i_)) * 64; \
                            _Pragma("unroll") for (int u = 0; u < TQ; ++u) rr[s_][u] = GV(seg_, u < nw ? u : nw - 1); \
                            xa_[s_] = GV(Xa, (i_)); xb_[s_] = GV(Xb, (i_)); ya_[s_] = GV(ysa, (i_)); yb_[s_] = GV(ysb, (i_)); }
#define MCX_Q2_FM(s_, i_) { double ya = ya_[s_], yb = yb_[s_]; const double xia = xa_[s_], xib = xb_[s_]; \
                            _Pragma("unroll") for (int u = 0; u < TQ; ++u) if (u < nw) { ya = dfma(rr[s_][u], xja[u], ya); yb = dfma(rr[s_][u], xjb[u], yb); } \
                            _Pragma("unroll") for (int u = 0; u < TQ; ++u) { \
                                Ya[u] = ((i_) == 0) ? rr[s_][u] * xia : dfma(rr[s_][u], xia, Ya[u]); \
                                Yb[u] = ((i_) == 0) ? rr[s_][u] * xib : dfma(rr[s_][u], xib, Yb[u]); } \
                            if (last) { qa = qa + ya * xia; qb = qb + yb * xib; } else { GV(ysa, (i_)) = ya; GV(ysb, (i_)) = yb; } }
#pragma unroll
            for (int s = 0; s < NB - 1; ++s) if (s < J0) MCX_Q2_LD(s, s)
            for (int i = 0; i < J0; i += NB) {
#pragma unroll
                for (int s = 0; s < NB; ++s) {
                    if (i + s + NB - 1 < J0) MCX_Q2_LD((s + NB - 1) % NB, i + s + NB - 1)
                    if (i + s < J0) MCX_Q2_FM(s, i + s)
                }
            }
#undef MCX_Q2_LD
#undef MCX_Q2_FM
        }
        // diagonal block: row i = J0 + ui takes its diagonal element on top of the finished column part, then the rest of its row
        {
            double da[TQ], db[TQ];
#define MCX_Q2_LDD(rv, i_) { const double *seg_ = St + (size_t)rowstart((i_), d) * 64; const int ui_ = (i_) - J0, m_ = d - 1 - (i_); \
                             _Pragma("unroll") for (int u = 0; u < TQ; ++u) { int k = u - ui_; k = k < 0 ? 0 : k; k = k > m_ ? m_ : k; rv[u] = GV(seg_, k); } }
#define MCX_Q2_FMD(rv, i_) { const int ui_ = (i_) - J0; double xia = 0.0, xib = 0.0, ya = 0.0, yb = 0.0; \
                             _Pragma("unroll") for (int u = 0; u < TQ; ++u) if (u == ui_) { xia = xja[u]; xib = xjb[u]; \
                                 ya = ((i_) == 0) ? rv[u] * xia : dfma(rv[u], xia, Ya[u]); yb = ((i_) == 0) ? rv[u] * xib : dfma(rv[u], \
                                     xib, Yb[u]); } \
                             _Pragma("unroll") for (int u = 0; u < TQ; ++u) if (u > ui_ && u < nw) { \
                                 ya = dfma(rv[u], xja[u], ya); yb = dfma(rv[u], xjb[u], yb); \
                                 Ya[u] = ((i_) == 0) ? rv[u] * xia : dfma(rv[u], xia, Ya[u]); \
                                 Yb[u] = ((i_) == 0) ? rv[u] * xib : dfma(rv[u], xib, Yb[u]); } \
                             if (last) { qa = qa + ya * xia; qb = qb + yb * xib; } else { GV(ysa, (i_)) = ya; GV(ysb, (i_)) = yb; } }
            MCX_Q2_LDD(da, J0)
            for (int i = J0; i < J0 + nw; i += 2) {
                if (i + 1 < J0 + nw) MCX_Q2_LDD(db, i + 1)
                MCX_Q2_FMD(da, i)
                if (i + 2 < J0 + nw) MCX_Q2_LDD(da, i + 2)
                if (i + 1 < J0 + nw) MCX_Q2_FMD(db, i + 1)
            }
#undef MCX_Q2_LDD
#undef MCX_Q2_FMD
        }
    }
}

template <bool LDSV>      // the two vectors in LDS (a compile-time choice: a pointer that is LDS or global at run time means FLAT accesses)
MCX_DEV void dr_body(const EngineDev &E, int it0, int it1, const double *__restrict__ g_mu, const double *__restrict__ g_lamT)
{
    extern __shared__ double X[];
    const int lane = threadIdx.x, tile = blockIdx.x, d = E.d;
    double *theta_t = E.theta + (size_t)tile * d * 64;
    double *c2_t = E.cs + (size_t)tile * 2 * d * 64;           // second-stage candidate (global scratch)
    // normals of the stage at hand, then dx_a = newpar2 - newpar  |  first-stage candidate, then dx_b = oldpar - newpar
    double *zb_t = LDSV ? X : c2_t + (size_t)d * 64;
    double *cand_t = LDSV ? X + (size_t)d * 64 : E.cand + (size_t)tile * d * 64;
    double *ysa_t = E.zs + (size_t)tile * 2 * d * 64, *ysb_t = ysa_t + (size_t)d * 64;     // row chains between two panels of iC
    const double *Rt = E.R + (size_t)tile * E.P * 64;

    Rng g;
    g.k0 = E.k0; g.k1 = E.chain_id0 + (uint32_t)(tile * 64 + lane);
    g.n = TIDX(E.rngn, tile, 1, 0, lane); g.cblk = 0; g.c2 = 0; g.c3 = 0;
    g.saved = (int)TIDX(E.ictr, tile, NICTR, I_SAVED, lane);
    g.saved_y = TIDX(E.scal, tile, NSCAL, S_SAVEDY, lane);
    double ss1 = TIDX(E.scal, tile, NSCAL, S_SS1, lane), pri1 = TIDX(E.scal, tile, NSCAL, S_PRI1, lane);
    double sigma2 = TIDX(E.scal, tile, NSCAL, S_SIGMA2, lane), alpha12 = TIDX(E.scal, tile, NSCAL, S_ALPHA12, lane);
    uint32_t stayed = TIDX(E.ictr, tile, NICTR, I_STAYED, lane), bnd = TIDX(E.ictr, tile, NICTR, I_BNDSTAYED, lane);
    uint32_t chainind = TIDX(E.ictr, tile, NICTR, I_CHAININD, lane), curcount = TIDX(E.ictr, tile, NICTR, I_CURCOUNT, lane);
    uint32_t dracc = TIDX(E.ictr, tile, NICTR, I_DRACC, lane), drtries = TIDX(E.ictr, tile, NICTR, I_DRTRIES, lane);

    for (int it = it0; it <= it1; ++it) {
        // ---- newpar = MCMC_propose(oldpar, R)   (the iteration's first draws: nothing else drew since the last one's end)
        gen_normals<MCX_RNG_NB>(g, zb_t, lane, d, true);
        if (E.usesvd) gemvN_panels(E.Rf + (size_t)tile * d * d * 64, zb_t, cand_t, theta_t, lane, d, true);   // matmulx(R,z)
        else trmv_panels<true>(Rt, zb_t, cand_t, theta_t, lane, d, true);
        // ---- bounds, prior, ss, alpha, reject
        bool inb = target_inbounds(E.tgt, d, lane, cand_t);
        double pri2 = target_prior(E.tgt, d, lane, cand_t);
        double ss2 = target_ss<false>(E.tgt, d, lane, cand_t, g_mu, g_lamT);
        bool reject;
        if (!inb) { reject = true; alpha12 = 0.0; }         // (with DR an out-of-bounds first stage is not counted, MCMC_run.F90:49)
        else {
            alpha12 = d_alpha(ss1, pri1, ss2, pri2, sigma2);
            reject = true;                                  // MCMC_reject, MCMC_DRAM.F90:140-155
            if (alpha12 >= 1.0) reject = false;
            else if (alpha12 > 0.0) { double u = rng_uniform(g); if (u <= alpha12) reject = false; }
        }
        // ---- second stage: one delayed-rejection try with R2 = R/drscale (MCMC_run.F90:65-91)
        bool dr_moved = false;
        if (__any(reject)) {
            const bool m = reject;
            if (m) drtries += 1;
            gen_normals<MCX_RNG_NB>(g, zb_t, lane, d, m);
            if (E.usesvd) gemvN_panels(E.R2f + (size_t)tile * d * d * 64, zb_t, c2_t, theta_t, lane, d, m);
            else trmv_panels<true>(E.R2 + (size_t)tile * E.P * 64, zb_t, c2_t, theta_t, lane, d, m);
            if (m) {
                bool inb2 = target_inbounds(E.tgt, d, lane, c2_t);
                if (!inb2) bnd += 1;
                else {
                    double pri3 = target_prior(E.tgt, d, lane, c2_t);
                    double ss3 = target_ss<false>(E.tgt, d, lane, c2_t, g_mu, g_lamT);
                    // MCMC_DR_alpha13, MCMC_DRAM.F90:162-186
                    double alpha32;
                    if (alpha12 == 0.0) alpha32 = 0.0;
                    else alpha32 = min1(d_exp(-0.5 * ((ss2 - ss3) / sigma2 + (pri2 - pri3))));
                    double l2 = -0.5 * ((ss3 - ss1) / sigma2 + (pri3 - pri1));
                    // dx_a = newpar2 - newpar, dx_b = oldpar - newpar take this lane's two LDS vectors (its normals and its
                    // first-stage candidate are dead from here on)
                    for (int k0 = 0; k0 < d; k0 += 8) {
                        double c1[8], c2[8], th[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) { const int k = (k0 + u < d) ? k0 + u : d - 1; c1[u] = GV(cand_t, k); c2[u] = GV(c2_t,
                            k); th[u] = GV(theta_t, k); }
#pragma unroll
                        for (int u = 0; u < 8; ++u) if (k0 + u < d) { GV(zb_t, k0 + u) = c2[u] - c1[u]; GV(cand_t, k0 + u) = th[u] - c1[u];
                            }
                    }
                    double qa, qb;
                    quadform2_panels(E.iC + (size_t)tile * E.P * 64, lane, d, zb_t, cand_t, ysa_t, ysb_t, qa, qb);
                    double q1 = -0.5 * (qa - qb);
                    double alpha13 = min1(d_exp(l2 + q1) * (1.0 - alpha32) / (1.0 - alpha12));
                    bool rej2 = true;
                    if (alpha13 >= 1.0) rej2 = false;
                    else if (alpha13 > 0.0) { double u = rng_uniform(g); if (u <= alpha13) rej2 = false; }
                    if (!rej2) { dracc += 1; reject = false; dr_moved = true; ss2 = ss3; pri2 = pri3; }
                }
            }
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
        if (!reject) {
            double *h = E.hist ? E.hist + ((size_t)tile * E.wcap + slot) * (size_t)E.hs * 64 : nullptr;
            if (dr_moved) copy_vec(theta_t, c2_t, h, lane, d);     // newpar = newpar2 when the DR try was accepted (two calls: a source
            else copy_vec(theta_t, cand_t, h, lane, d);            // that is global or LDS by the lane would mean FLAT accesses)
            if (h) GV(h, d) = ss1;
        }
        if (E.hist) {
            if (lane == 0) E.wacc[(size_t)tile * E.wcap + slot] = ballot;
            if (E.record_s2) E.s2hist[((size_t)tile * E.wcap + slot) * 64 + lane] = sigma2;
        }
        if (E.accmask && lane == 0) E.accmask[(size_t)(it - 1) * E.ntiles + tile] = ballot;
    }

    TIDX(E.rngn, tile, 1, 0, lane) = g.n;
    TIDX(E.ictr, tile, NICTR, I_SAVED, lane) = (uint32_t)g.saved;
    TIDX(E.scal, tile, NSCAL, S_SAVEDY, lane) = g.saved_y;
    TIDX(E.scal, tile, NSCAL, S_SS1, lane) = ss1; TIDX(E.scal, tile, NSCAL, S_PRI1, lane) = pri1;
    TIDX(E.scal, tile, NSCAL, S_SIGMA2, lane) = sigma2; TIDX(E.scal, tile, NSCAL, S_ALPHA12, lane) = alpha12;
    TIDX(E.ictr, tile, NICTR, I_STAYED, lane) = stayed; TIDX(E.ictr, tile, NICTR, I_BNDSTAYED, lane) = bnd;
    TIDX(E.ictr, tile, NICTR, I_CHAININD, lane) = chainind; TIDX(E.ictr, tile, NICTR, I_CURCOUNT, lane) = curcount;
    TIDX(E.ictr, tile, NICTR, I_DRACC, lane) = dracc; TIDX(E.ictr, tile, NICTR, I_DRTRIES, lane) = drtries;
}
__global__ __launch_bounds__(64, 2) void step_kernel_dr(EngineDev E, int it0, int it1, const double *__restrict__ g_mu,
    const double *__restrict__ g_lamT)
{ dr_body<true>(E, it0, it1, g_mu, g_lamT); }
// npar > 160: the same with the two vectors in global scratch
__global__ __launch_bounds__(64, 2) void step_kernel_dr_big(EngineDev E, int it0, int it1, const double *__restrict__ g_mu,
    const double *__restrict__ g_lamT)
{ dr_body<false>(E, it0, it1, g_mu, g_lamT); }

#ifndef MCX_AM_WAVES
#define MCX_AM_WAVES 2
#endif
#ifndef MCX_AM_WIDE
#define MCX_AM_WIDE true
#endif
template <bool RAM, bool DR, bool POOLED>
__global__ __launch_bounds__(64, RAM ? MCX_RAM_WAVES : (DR || POOLED) ? 2 : MCX_AM_WAVES) void step_kernel(EngineDev E, int it0, int it1,
    const double *__restrict__ ramscale,
                                                     const double *__restrict__ g_mu, const double *__restrict__ g_lamT,
                                                     const double *__restrict__ g_sharedR)
{ step_body<RAM, DR, POOLED, (RAM || (!DR && !POOLED && MCX_AM_WIDE))>(E, it0, it1, ramscale, g_mu, g_lamT, g_sharedR); }
// the plain AM / Metropolis / ER step with the state vector, the candidate and the two normal vectors in LDS (EngineDev::lds_scratch)
__global__ __launch_bounds__(64, MCX_AM_WAVES) void step_kernel_ldsv(EngineDev E, int it0, int it1, const double *__restrict__ ramscale,
                                                     const double *__restrict__ g_mu, const double *__restrict__ g_lamT,
                                                     const double *__restrict__ g_sharedR)
{ step_body<false, false, false, MCX_AM_WIDE, false, true>(E, it0, it1, ramscale, g_mu, g_lamT, g_sharedR); }
// ... and the chain's packed factor too (npar <= TW; EngineDev::lds_scratch == 2): north_star's "Cholesky factor staged in LDS"
__global__ __launch_bounds__(64, MCX_AM_WAVES) void step_kernel_ldsr(EngineDev E, int it0, int it1, const double *__restrict__ ramscale,
                                                     const double *__restrict__ g_mu, const double *__restrict__ g_lamT,
                                                     const double *__restrict__ g_sharedR)
{ step_body<false, false, false, MCX_AM_WIDE, false, true, true>(E, it0, it1, ramscale, g_mu, g_lamT, g_sharedR); }

// method='ram' at npar <= RW with few enough tiles: the factor in LDS for the launch (EngineDev::lds_scratch == 3)
__global__ __launch_bounds__(64, 2) void step_kernel_ram_ldsr(EngineDev E, int it0, int it1, const double *__restrict__ ramscale,
                                                     const double *__restrict__ g_mu, const double *__restrict__ g_lamT,
                                                     const double *__restrict__ g_sharedR)
{ step_body<true, false, false, true, false, false, true>(E, it0, it1, ramscale, g_mu, g_lamT, g_sharedR); }

// method='ram' above npar 20: step_kernel<true, false, false> with the wide column panels (RW_WIDE above)
__global__ __launch_bounds__(64, MCX_RAM_WAVES) void step_kernel_ram_wide(EngineDev E, int it0, int it1,
    const double *__restrict__ ramscale,
                                                     const double *__restrict__ g_mu, const double *__restrict__ g_lamT,
                                                     const double *__restrict__ g_sharedR)
{ step_body<true, false, false, true, false, false, false, false, RW_WIDE>(E, it0, it1, ramscale, g_mu, g_lamT, g_sharedR); }

// pooled mode with delayed rejection: the shared factor, its second-stage copy R2 = R / drscale and the shared inverse
// covariance iC = dpotri(R) all come through the scalar cache (the host recomputes the three at every pooled tick)
__global__ __launch_bounds__(64, 2) void step_kernel_pooled_dr(EngineDev E, int it0, int it1, const double *__restrict__ ramscale,
                                                               const double *__restrict__ g_mu, const double *__restrict__ g_lamT,
                                                               const double *__restrict__ g_sharedR, const double *__restrict__ g_sharedR2,
                                                               const double *__restrict__ g_sharediC)
{ step_body<false, true, true, false>(E, it0, it1, ramscale, g_mu, g_lamT, g_sharedR, g_sharedR2, g_sharediC); }
// ... with the two quadratic-form vectors in global scratch (EngineDev::xscr): above npar 20 the LDS form costs waves (51 KiB per wave
// at npar 50: three waves per CU), and above 160 it does not fit at all
__global__ __launch_bounds__(64, 2) void step_kernel_pooled_dr_big(EngineDev E, int it0, int it1, const double *__restrict__ ramscale,
                                                                   const double *__restrict__ g_mu, const double *__restrict__ g_lamT,
                                                                   const double *__restrict__ g_sharedR,
                                                                       const double *__restrict__ g_sharedR2,
                                                                   const double *__restrict__ g_sharediC)
{ step_body<false, true, true, false, false, false, false, true>(E, it0, it1, ramscale, g_mu, g_lamT, g_sharedR, g_sharedR2, g_sharediC); }

// method='ram' with condmax > 0: the factor is the full SVD one (E.Rf), proposals are matmulx(R,u), the rank-one
// adaptation runs on its upper triangle (ram_update_full)
__global__ __launch_bounds__(64, 2) void step_kernel_ram_fullr(EngineDev E, int it0, int it1, const double *__restrict__ ramscale,
                                                               const double *__restrict__ g_mu, const double *__restrict__ g_lamT,
                                                               const double *__restrict__ g_sharedR)
{ step_body<true, false, false, false, true>(E, it0, it1, ramscale, g_mu, g_lamT, g_sharedR); }

} // namespace mcx
