// mcx_phase.hpp -- the iteration cut at the user's evaluations: host_phase_kernel (host callbacks), dev_eval_kernel + step_kernel_cols
// (device target with response columns, nycol >= 1, in one launch), run1_kernel (MCMC_run1 / MCMC_run1_er)
// (one of the family headers mcx_kernels.hpp includes, in this order: mcx_common, mcx_products, mcx_step, mcx_scam, mcx_pooled, mcx_phase,
// mcx_adapt, mcx_svd, mcx_moments)
#pragma once
#include "mcx_pooled.hpp"

namespace mcx {

// ---------------------------------------------------------------- nycol > 1: sums over the response columns
// sum((a - b)/sigma2) and friends reduce from 0 in column order, like the reference's array expressions
// (MCMC_DRAM.F90:111, 129, 176-179); a, b, s2 are per-chain vectors (element j at GV(p, j)).
MCX_DEV double colsum_diff(const double *a, const double *b, const double *s2, int ny, int lane)
{
    double s = 0.0;
    for (int j = 0; j < ny; ++j) s = s + (GV(a, j) - GV(b, j)) / GV(s2, j);
    return s;
}
MCX_DEV double d_alpha_cols(const double *ss1, double pri1, const double *ss2, double pri2, const double *s2, int ny, int lane)
{
    double tst = -0.5 * (colsum_diff(ss2, ss1, s2, ny, lane) + (pri2 - pri1));
    double a;
    if (tst >= 0.0) a = 1.0;
    else if (tst < -708.39641853226408) a = 0.0;
    else a = d_exp(tst);
    return a;
}

// end of an iteration: MCMC_run.F90:93-105 / MCMC_run_ram.F90:66-78
// ss2cols: with nycol > 1 the accepted point's ss per column (a per-chain vector); nullptr = the scalar ss2
// MSEL: the method class at compile time -- -1: whatever the engine holds (the host-callback phase kernels), 0: not RAM, 1: RAM, 2: SCAM
// (step_kernel_cols: an instantiation without the RAM update's panels needs a third fewer registers)
template <int MSEL = -1>
MCX_DEV void host_finish(const EngineDev &E, int tile, int lane, int it, LaneState &L, bool reject, bool dr_moved,
                         double ss2, double pri2, const double *ramscale, const double *ss2cols = nullptr)
{
    const int d = E.d, ny = E.ny;
    double *ssv = ny > 1 ? E.ssv + (size_t)tile * ny * 64 : nullptr, *s2v = ny > 1 ? E.s2v + (size_t)tile * ny * 64 : nullptr;
    double *theta_t = E.theta + (size_t)tile * d * 64;
    double *cand_t = E.cand + (size_t)tile * d * 64;
    double *zs_t = E.zs + (size_t)tile * 2 * d * 64;
    double *cs_t = E.cs + (size_t)tile * 2 * d * 64;
    if (reject) { L.stayed += 1; L.curcount += 1; }
    else {
        if (ny > 1) { for (int j = 0; j < ny; ++j) GV(ssv, j) = GV(ss2cols, j); ss2 = GV(ssv, 0); }
        L.ss1 = ss2; L.pri1 = pri2; L.chainind += 1; L.curcount = 1;
    }
    if (E.updatesigma) {                                // MCMC_updatesigma2: one gamma draw per column, in column order
        if (ny > 1) {
            for (int j = 0; j < ny; ++j) {
                double gm = rng_gamma(L.g, E.gshapev[j], 2.0 / (E.N0S02 + GV(ssv, j)));
                GV(s2v, j) = 1.0 / gm;
            }
            L.sigma2 = GV(s2v, 0);
        } else {
            double gm = rng_gamma(L.g, E.gam_shape, 2.0 / (E.N0S02 + L.ss1));
            L.sigma2 = 1.0 / gm;
        }
    }
    unsigned long long ballot = __ballot(!reject);
    const int slot = it % E.wcap;
    if (!reject) {
        double *h = E.hist ? E.hist + ((size_t)tile * E.wcap + slot) * (size_t)E.hs * 64 : nullptr;
        const double *src = dr_moved ? cs_t : cand_t;
        copy_vec(theta_t, src, h, lane, d);
        if (h) { GV(h, d) = L.ss1; for (int j = 1; j < ny; ++j) GV(h, d + j) = GV(ssv, j); }
    }
    if (E.hist) {
        if (lane == 0) E.wacc[(size_t)tile * E.wcap + slot] = ballot;
        if (E.record_s2) {
            if (ny > 1) { for (int j = 0; j < ny; ++j) E.s2hist[(((size_t)tile * E.wcap + slot) * ny + j) * 64 + lane] = GV(s2v, j); }
            else E.s2hist[((size_t)tile * E.wcap + slot) * 64 + lane] = L.sigma2;
        }
    }
    if (E.accmask && lane == 0) E.accmask[(size_t)(it - 1) * E.ntiles + tile] = ballot;
    if ((MSEL < 0 ? E.method == M_RAM : MSEL == 1) && E.doadapt != 0 && !(it < E.burnintime && E.doburnin != 0)) {
        double a = ramscale[0] * (L.alpha12 - E.alphatarget);
        if (!(a >= 0.0)) TIDX(E.ictr, tile, NICTR, I_DOWNS, lane) += 1u;
        const double *hx = E.hx + (size_t)tile * NHX * 64;
        if (E.usesvd) ram_update_full(E.Rf + (size_t)tile * d * d * 64, zs_t, cs_t, lane, d, a, GV(hx, HX_SU), true, L.status);
        else { bool pd = L.pdesc != 0u; ram_update<false>(E.R + (size_t)tile * E.P * 64, zs_t, zs_t, cs_t, cand_t, theta_t, lane, d, a,
            GV(hx, HX_SU), true, false, L.status, nullptr, pd); L.pdesc = pd ? 1u : 0u; }
    }
}

// Device-resident evaluation for the phase-cut iteration: fills hev (inbounds, prior, ss per response column) the way
// host_eval does from the user's host callbacks, for the built-in response-column target
//   ss_j(theta) = sum_i (y_j(i) - theta_1 exp(-theta_{1+j} x_i))**2,  j = 1..nycol  (oracle/mcx_targets.h: mcxt_ss_expdata_cols)
// with the library's box bounds and Gaussian priors.  what: 0 = checkbounds, priorfun, ssfunction; 1 = checkbounds and
// priorfun; 2 = ssfunction alone (ssfunction_er0.f90: the default ssfunction_er is ssfunction).
MCX_DEV void dev_eval_body(const EngineDev &E, int tile, int lane, const double *src, int stride_k, int use_stage2, int what)
{
    const int d = E.d, ny = E.ny;
    const double *c_t = src + (size_t)tile * stride_k * 64;
    double *hev = E.hev + (size_t)tile * (NHE - 1 + ny) * 64;
    const double *hx = E.hx + (size_t)tile * NHX * 64;
    bool inb = true;
    double pri = 0.0;
    const bool skip = use_stage2 && GV(hx, HX_STAGE2) == 0.0;
    if (!skip && what != 2) {
        inb = target_inbounds(E.tgt, d, lane, c_t);
        if (inb) pri = target_prior(E.tgt, d, lane, c_t);          // MCMC_run.F90:54-56: prior first
    }
    const bool doss = !skip && ((what == 0 && inb) || what == 2);
    const double th0 = GV(c_t, 0);
    for (int j = 0; j < ny; ++j) {
        double ss = 0.0;
        if (doss) {
            const double thj = GV(c_t, 1 + j);
            const double *yj = E.tgt.y + (size_t)j * E.tgt.ndata;
            for (int i = 0; i < E.tgt.ndata; ++i) {
                double r = yj[i] - th0 * d_exp(-(thj * E.tgt.x[i]));
                ss = dfma(r, r, ss);
            }
        }
        GV(hev, HE_SS + j) = ss;
    }
    GV(hev, HE_INB) = (inb && !skip) ? 1.0 : 0.0;
    GV(hev, HE_PRI) = pri;
}
__global__ __launch_bounds__(64) void dev_eval_kernel(EngineDev E, const double *__restrict__ src, int stride_k, int use_stage2, int what)
{ dev_eval_body(E, blockIdx.x, threadIdx.x, src, stride_k, use_stage2, what); }

// sR / sR2 / siC: pooled mode's shared factor, second-stage factor and inverse covariance (nullptr: the chain's own)
template <int PHASE, int MSEL = -1>
MCX_DEV void host_phase_body(const EngineDev &E, int tile, int lane, int it, const double *__restrict__ ramscale, int aux, double *X,
                             const double *__restrict__ sR = nullptr, const double *__restrict__ sR2 = nullptr,
                                 const double *__restrict__ siC = nullptr)
{
    const int d = E.d;
    double *theta_t = E.theta + (size_t)tile * d * 64;
    double *cand_t = E.cand + (size_t)tile * d * 64;
    double *zs_t = E.zs + (size_t)tile * 2 * d * 64;           // host mode: first half = stage-1 z, second half = stage-2 z
    double *c2_t = E.cs + (size_t)tile * 2 * d * 64;
    const int ny = E.ny;
    double *hev = E.hev + (size_t)tile * (NHE - 1 + ny) * 64;    // inbounds, prior, ss per column
    double *hx = E.hx + (size_t)tile * NHX * 64;
    double *Y = X + (size_t)d * 64;
    double *ssv = ny > 1 ? E.ssv + (size_t)tile * ny * 64 : nullptr, *s2v = ny > 1 ? E.s2v + (size_t)tile * ny * 64 : nullptr;
    double *ss2v = ny > 1 ? E.ss2v + (size_t)tile * ny * 64 : nullptr;
    const double *sshev = hev + (size_t)HE_SS * 64;              // the host's ss columns of the point just evaluated
    LaneState L;
    lane_load(E, tile, lane, L);
    if (PHASE == 0) {                                             // newpar = MCMC_propose(oldpar, R)
        double su = gen_normals(L.g, zs_t, lane, d, true);
        GV(hx, HX_SU) = su;
        if (sR) { if (E.usesvd) gemvN_shared(sR, zs_t, cand_t, theta_t, lane, d); else trmv_shared(sR, zs_t, cand_t, theta_t, lane, d); }
        else if (E.usesvd) gemvN_panels(E.Rf + (size_t)tile * d * d * 64, zs_t, cand_t, theta_t, lane, d, true);    // matmulx(R,z)
        else trmv_panels<false>(E.R + (size_t)tile * E.P * 64, zs_t, cand_t, theta_t, lane, d, true, E.method == M_RAM && L.pdesc != 0u);
    } else if (PHASE == 1) {
        const bool inb = GV(hev, HE_INB) != 0.0;
        const double pri2 = GV(hev, HE_PRI), ss2 = GV(hev, HE_SS);
        bool reject;
        if (!inb) {
            if (!E.dodr) L.bnd += 1;
            reject = true;
            if (E.method != M_RAM) L.alpha12 = 0.0;
        } else {
            L.alpha12 = ny > 1 ? d_alpha_cols(ssv, L.pri1, sshev, pri2, s2v, ny, lane) : d_alpha(L.ss1, L.pri1, ss2, pri2, L.sigma2);
            reject = true;
            if (L.alpha12 >= 1.0) reject = false;
            else if (L.alpha12 > 0.0) { double u = rng_uniform(L.g); if (u <= L.alpha12) reject = false; }
        }
        if (E.dodr) {
            const bool m = reject;
            if (m) L.drtries += 1;
            for (int j = 0; j < (ny > 1 ? ny : 0); ++j) GV(ss2v, j) = GV(sshev, j);
            gen_normals(L.g, zs_t + (size_t)d * 64, lane, d, m);
            if (sR2) { if (E.usesvd) gemvN_shared(sR2, zs_t + (size_t)d * 64, c2_t, theta_t, lane, d); else trmv_shared(sR2,
                zs_t + (size_t)d * 64, c2_t, theta_t, lane, d); }
            else if (E.usesvd) gemvN_panels(E.R2f + (size_t)tile * d * d * 64, zs_t + (size_t)d * 64, c2_t, theta_t, lane, d, m);
            else trmv_panels<false>(E.R2 + (size_t)tile * E.P * 64, zs_t + (size_t)d * 64, c2_t, theta_t, lane, d, m);
            GV(hx, HX_SS2) = ss2; GV(hx, HX_PRI2) = pri2;
            GV(hx, HX_REJECT) = reject ? 1.0 : 0.0; GV(hx, HX_STAGE2) = m ? 1.0 : 0.0;
        } else {
            host_finish<MSEL>(E, tile, lane, it, L, reject, false, ss2, pri2, ramscale, sshev);
        }
    } else if (PHASE == 5) {                                      // SCAM sub-step aux: propose (MCMC_run_scam.F90:94-117)
        const int j = aux;
        double *rot_t = c2_t;
        const double *Ut = E.Rf + (size_t)tile * d * d * 64;
        if (j == 0) GV(hx, HX_MOVED) = 0.0;
        if (E.scam_fast) {
            const double zj = rng_normal(L.g) * TIDX(E.qstd, tile, d, j, lane);
            scam_fast_propose(Ut, theta_t, cand_t, lane, d, j, zj);
        } else {
            gemvT_panels(Ut, theta_t, rot_t, lane, d);
            const double zj = rng_normal(L.g) * TIDX(E.qstd, tile, d, j, lane);
            GV(rot_t, j) = GV(rot_t, j) + zj;
            gemvN_panels(Ut, rot_t, cand_t, nullptr, lane, d, true);
        }
    } else if (PHASE == 6) {                                      // SCAM sub-step: decide with the host's bounds / prior / ss
        const bool inb = GV(hev, HE_INB) != 0.0;
        const double pri2 = GV(hev, HE_PRI), ss2 = GV(hev, HE_SS);
        bool reject;
        if (!inb) { L.bnd += 1; L.alpha12 = 0.0; reject = true; }
        else {
            L.alpha12 = ny > 1 ? d_alpha_cols(ssv, L.pri1, sshev, pri2, s2v, ny, lane) : d_alpha(L.ss1, L.pri1, ss2, pri2, L.sigma2);
            reject = true;
            if (L.alpha12 >= 1.0) reject = false;
            else if (L.alpha12 > 0.0) { double u = rng_uniform(L.g); if (u <= L.alpha12) reject = false; }
        }
        if (!reject) {
            for (int j = 0; j < (ny > 1 ? ny : 0); ++j) GV(ssv, j) = GV(sshev, j);
            L.ss1 = ss2; L.pri1 = pri2; GV(hx, HX_MOVED) = 1.0;
            copy_vec(theta_t, cand_t, nullptr, lane, d);
        }
    } else if (PHASE == 7) {                                      // SCAM: end of the outer iteration (one chain row)
        const bool rejall = GV(hx, HX_MOVED) == 0.0;
        if (rejall) { L.stayed += 1; L.curcount += 1; }
        else { L.chainind += 1; L.curcount = 1; }
        if (E.updatesigma) {
            if (ny > 1) {
                for (int j = 0; j < ny; ++j) { double gm = rng_gamma(L.g, E.gshapev[j], 2.0 / (E.N0S02 + GV(ssv, j))); GV(s2v,
                    j) = 1.0 / gm; }
                L.sigma2 = GV(s2v, 0);
            } else {
                double gm = rng_gamma(L.g, E.gam_shape, 2.0 / (E.N0S02 + L.ss1));
                L.sigma2 = 1.0 / gm;
            }
        }
        unsigned long long ballot = __ballot(!rejall);
        const int slot = it % E.wcap;
        if (E.hist) {
            if (!rejall) {
                double *h = E.hist + ((size_t)tile * E.wcap + slot) * (size_t)E.hs * 64;
                for (int k = 0; k < d; ++k) GV(h, k) = GV(theta_t, k);
                GV(h, d) = L.ss1;
                for (int j = 1; j < ny; ++j) GV(h, d + j) = GV(ssv, j);
            }
            if (lane == 0) E.wacc[(size_t)tile * E.wcap + slot] = ballot;
            if (E.record_s2) {
                if (ny > 1) { for (int j = 0; j < ny; ++j) E.s2hist[(((size_t)tile * E.wcap + slot) * ny + j) * 64 + lane] = GV(s2v, j); }
                else E.s2hist[((size_t)tile * E.wcap + slot) * 64 + lane] = L.sigma2;
            }
        }
        if (E.accmask && lane == 0) E.accmask[(size_t)(it - 1) * E.ntiles + tile] = ballot;
    } else if (PHASE == 3) {                                      // early rejection, first half (MCMC_run_er.F90:54-70)
        // the host has evaluated checkbounds and priorfun; draw the threshold, test the prior, leave sscrit for ssfunction_er
        const bool inb = GV(hev, HE_INB) != 0.0;
        const double pri2 = GV(hev, HE_PRI);
        bool reject = false, need = false;
        double crit = 0.0;
        if (!inb) { L.bnd += 1; reject = true; }
        else {
            double u = rng_uniform(L.g);                          // MCMC_sscrit, MCMC_DRAM.F90:124-135: always drawn
            double s1 = L.ss1 / L.sigma2;
            if (ny > 1) { s1 = 0.0; for (int j = 0; j < ny; ++j) s1 = s1 + GV(ssv, j) / GV(s2v, j); }      // sum(ss1/sigma2)
            double sscrit = -2.0 * d_log(u) + s1 + L.pri1;
            if (pri2 >= sscrit) { reject = true; TIDX(E.ictr, tile, NICTR, I_ERSTAYED, lane) += 1; }
            else { crit = L.sigma2 * (sscrit - pri2); need = true; }    // sigma2(1): MCMC_run_er.F90:72
        }
        GV(hx, HX_PRI2) = pri2; GV(hx, HX_CRIT) = crit;
        GV(hx, HX_REJECT) = reject ? 1.0 : 0.0; GV(hx, HX_STAGE2) = need ? 1.0 : 0.0;
    } else if (PHASE == 4) {                                      // early rejection, second half (:71-101)
        bool reject = GV(hx, HX_REJECT) != 0.0;
        const double pri2 = GV(hx, HX_PRI2);
        double ss2 = 0.0;
        if (GV(hx, HX_STAGE2) != 0.0) {
            ss2 = GV(hev, HE_SS);
            double tot = ss2;
            if (ny > 1) { tot = 0.0; for (int j = 0; j < ny; ++j) tot = tot + GV(sshev, j); }                 // sum(ss2)
            reject = (tot >= GV(hx, HX_CRIT));
        }
        host_finish<MSEL>(E, tile, lane, it, L, reject, false, ss2, pri2, ramscale, sshev);
    } else {                                                      // PHASE 2: decide the DR try, finish
        bool reject = GV(hx, HX_REJECT) != 0.0;
        double ss2 = GV(hx, HX_SS2), pri2 = GV(hx, HX_PRI2);
        bool dr_moved = false;
        if (GV(hx, HX_STAGE2) != 0.0) {
            const bool inb2 = GV(hev, HE_INB) != 0.0;
            if (!inb2) L.bnd += 1;
            else {
                const double pri3 = GV(hev, HE_PRI), ss3 = GV(hev, HE_SS);
                double alpha32, l2;
                if (ny > 1) {
                    if (L.alpha12 == 0.0) alpha32 = 0.0;
                    else alpha32 = min1(d_exp(-0.5 * (colsum_diff(ss2v, sshev, s2v, ny, lane) + (pri2 - pri3))));
                    l2 = -0.5 * (colsum_diff(sshev, ssv, s2v, ny, lane) + (pri3 - L.pri1));
                } else {
                    if (L.alpha12 == 0.0) alpha32 = 0.0;
                    else alpha32 = min1(d_exp(-0.5 * ((ss2 - ss3) / L.sigma2 + (pri2 - pri3))));
                    l2 = -0.5 * ((ss3 - L.ss1) / L.sigma2 + (pri3 - L.pri1));
                }
                const double *iCt = siC ? nullptr : E.iC + (size_t)tile * E.P * 64;
                double *Xq = E.dr_lds ? X : zs_t, *Yq = E.dr_lds ? Y : zs_t + (size_t)d * 64;   // npar > 160: the (dead) normal vectors
                for (int k = 0; k < d; ++k) GV(Xq, k) = GV(c2_t, k) - GV(cand_t, k);
                double qa = siC ? quadform_sym_shared(siC, lane, d, Xq, Yq) : quadform_sym(iCt, lane, d, Xq, Yq);
                for (int k = 0; k < d; ++k) GV(Xq, k) = GV(theta_t, k) - GV(cand_t, k);
                double qb = siC ? quadform_sym_shared(siC, lane, d, Xq, Yq) : quadform_sym(iCt, lane, d, Xq, Yq);
                double q1 = -0.5 * (qa - qb);
                double alpha13 = min1(d_exp(l2 + q1) * (1.0 - alpha32) / (1.0 - L.alpha12));
                bool rej2 = true;
                if (alpha13 >= 1.0) rej2 = false;
                else if (alpha13 > 0.0) { double u = rng_uniform(L.g); if (u <= alpha13) rej2 = false; }
                if (!rej2) { L.dracc += 1; reject = false; dr_moved = true; ss2 = ss3; pri2 = pri3; }
            }
        }
        host_finish<MSEL>(E, tile, lane, it, L, reject, dr_moved, ss2, pri2, ramscale, dr_moved ? sshev : ss2v);
    }
    lane_store(E, tile, lane, L);
}
template <int PHASE>
__global__ __launch_bounds__(64) void host_phase_kernel(EngineDev E, int it, const double *__restrict__ ramscale, int aux)
{
    extern __shared__ double X[];
    host_phase_body<PHASE>(E, blockIdx.x, threadIdx.x, it, ramscale, aux, X);
}

// Two (three) phases that no evaluation separates, in one launch: the last phase of an iteration and the first of the next one (the
// proposal), a SCAM sub-step's decision and the next component's proposal.  With the user's functions on the host an iteration of few
// chains is launch and wake-up latency, nothing else -- one launch per evaluation instead of two.  The phases hand over through the chain's
// own state exactly as separate launches do (every element written and read back by the same lane: see step_kernel_cols).  PB / PC < 0:
// none.
template <int PA, int PB, int PC>
__global__ __launch_bounds__(64) void host_phase_seq_kernel(EngineDev E, int itA, int auxA, int itB, int auxB, int itC, int auxC,
    const double *__restrict__ ramscale)
{
    extern __shared__ double X[];
    host_phase_body<PA>(E, blockIdx.x, threadIdx.x, itA, ramscale + itA, auxA, X);
    if constexpr (PB >= 0) host_phase_body<PB>(E, blockIdx.x, threadIdx.x, itB, ramscale + itB, auxB, X);
    if constexpr (PC >= 0) host_phase_body<PC>(E, blockIdx.x, threadIdx.x, itC, ramscale + itC, auxC, X);
}

// ---------------------------------------------------------------- nycol > 1 in ONE launch (step_kernel_cols)
// Iterations it0..it1 of MCMC_run / MCMC_run_ram / MCMC_run_er / MCMC_run_scam for a target the DEVICE evaluates between the phases
// of an iteration (the response-column target `expcols`: nycol sums of squares per point, one sigma2 per column, sums over the
// columns in MCMC_alpha, MCMC_sscrit and MCMC_DR_alpha13, one gamma draw per column -- MCMC_DRAM.F90:100-135,162-206): the phase
// bodies of the host-callback path and dev_eval_body in the order host_iteration launches them, fused into one kernel.  The phases
// hand their intermediate results over through the chain's own global scratch (hev, hx, cand, ...) exactly as the separate launches
// do -- every element is written and read back by the same lane, so program order is all the ordering there is to keep -- which makes
// the fused form the phase form bit for bit (tests/test_gpu_host_callbacks.py, fixtures m1..m5 both ways).  ramscale: the table's
// base (1 / it**nuparam at index it).  sR / sR2 / siC: pooled mode's shared tables.
// MSEL: 0 = MCMC_run / MCMC_run_er (with or without delayed rejection), 1 = MCMC_run_ram, 2 = MCMC_run_scam -- one instantiation each, so that
// none carries the others' code (the one general kernel of round 5 held 294 registers and 506 spilled SGPRs and ran one wave per SIMD; two
// waves: config 1's model with two response columns 2.26 -> 3.09e9 proposals/s even with that kernel, profiles/r06_e/cols_waves.txt)
#ifndef MCX_COLS_WAVES
#define MCX_COLS_WAVES 2
#endif
template <int MSEL>
__global__ __launch_bounds__(64, MCX_COLS_WAVES) void step_kernel_cols(EngineDev E, int it0, int it1, const double *__restrict__ ramscale,
                                                                       const double *__restrict__ sR, const double *__restrict__ sR2,
                                                                       const double *__restrict__ siC)
{
    extern __shared__ double X[];
    const int lane = threadIdx.x, tile = blockIdx.x, d = E.d;
    for (int it = it0; it <= it1; ++it) {
        const double *rs = ramscale + it;
        if constexpr (MSEL == 2) {                      // MCMC_run_scam.F90:94-138: npar componentwise proposals, each with its own evaluation
            for (int j = 0; j < d; ++j) {
                host_phase_body<5, MSEL>(E, tile, lane, it, rs, j, X);
                dev_eval_body(E, tile, lane, E.cand, d, 0, 0);
                host_phase_body<6, MSEL>(E, tile, lane, it, rs, j, X);
            }
            host_phase_body<7, MSEL>(E, tile, lane, it, rs, 0, X);
        } else {
            host_phase_body<0, MSEL>(E, tile, lane, it, rs, 0, X, sR, sR2, siC);
            if (MSEL == 0 && E.method == M_ER) {        // MCMC_run_er.F90:54-101: the threshold is drawn between priorfun and ssfunction
                dev_eval_body(E, tile, lane, E.cand, d, 0, 1);
                host_phase_body<3, MSEL>(E, tile, lane, it, rs, 0, X, sR, sR2, siC);
                dev_eval_body(E, tile, lane, E.cand, d, 1, 2);
                host_phase_body<4, MSEL>(E, tile, lane, it, rs, 0, X, sR, sR2, siC);
                continue;
            }
            dev_eval_body(E, tile, lane, E.cand, d, 0, 0);
            host_phase_body<1, MSEL>(E, tile, lane, it, rs, 0, X, sR, sR2, siC);
            if (MSEL == 0 && E.dodr) {
                dev_eval_body(E, tile, lane, E.cs, 2 * d, 1, 0);
                host_phase_body<2, MSEL>(E, tile, lane, it, rs, 0, X, sR, sR2, siC);
            }
        }
    }
    // pooled method = 'ram': the tick's statistic reads the last iteration's normals where the single-launch kernels leave them,
    // in the (it & 1) half of the chain's two normal vectors (moments_kernel kind 2); the phases keep stage-1 normals in the first half
    if (MSEL != 2 && sR && !E.dodr && (it1 & 1)) {
        double *zs_t = E.zs + (size_t)tile * 2 * d * 64;
        for (int k = 0; k < d; ++k) GV(zs_t, d + k) = GV(zs_t, k);
    }
}

// ---------------------------------------------------------------- MCMC_run1 / MCMC_run1_er: one evaluation per invocation
// The reference's file protocol (MCMC_run1.F90:31-256, MCMC_run1_er.F90:28-234) keeps the chain's state in files between
// program runs; what is arithmetic in it -- the acceptance probability of the point just evaluated, MCMC_reject's draw,
// the next proposal, early rejection's threshold -- runs here, on the engine's factors (R, R2, iC of mcmcx_init) and
// the chain's stream.  The caller's vectors travel in r1, tile-interleaved like everything else:
//   [0,d) the current point (oldpar2; `from` of a proposal)   [d,2d) oldpar1   [2d,3d) newpar (a proposal's result)
//   then ny each: ssprev2, ssprev1, ss;  then the scalars below.
enum { R1_PRI2 = 0, R1_PRI1, R1_PRI, R1_A12, R1_ALPHA, R1_REJECT, R1_CRIT, R1_SPARE, NR1 };
MCX_DEV int r1_len(int d, int ny) { return 3 * d + 3 * ny + NR1; }
// MODE 0: alpha = MCMC_alpha(oldpar1 -> newpar) (drstage 1, MCMC_run1.F90:141) or MCMC_DR_alpha13(oldpar2, oldpar1,
//         newpar) (drstage 2, :137-139), then MCMC_reject(alpha) (:143)
// MODE 1 / 2: newpar = MCMC_propose(from, R) / (from, R2)  (:185-189)
// MODE 3: sscrit = MCMC_sscrit(ssprev1, sspri1) (MCMC_run1_er.F90:168; MCMC_DRAM.F90:124-135)
template <int MODE>
__global__ __launch_bounds__(64) void run1_kernel(EngineDev E, double *r1, int drstage)
{
    extern __shared__ double X[];
    const int lane = threadIdx.x, tile = blockIdx.x, d = E.d, ny = E.ny;
    double *b = r1 + (size_t)tile * r1_len(d, ny) * 64;
    double *cur_t = b, *old1_t = b + (size_t)d * 64, *new_t = b + (size_t)2 * d * 64;
    double *ssp2 = b + (size_t)3 * d * 64, *ssp1 = ssp2 + (size_t)ny * 64, *ssn = ssp1 + (size_t)ny * 64;
    double *sc = ssn + (size_t)ny * 64;
    const double *s2v = ny > 1 ? E.s2v + (size_t)tile * ny * 64 : nullptr;
    double *zs_t = E.zs + (size_t)tile * 2 * d * 64;
    double *Y = X + (size_t)d * 64;
    LaneState L;
    lane_load(E, tile, lane, L);
    if (MODE == 0) {
        const double pri1 = GV(sc, R1_PRI1), pri = GV(sc, R1_PRI);
        double alpha;
        if (drstage > 1 && E.dodr) {                      // MCMC_DR_alpha13, MCMC_DRAM.F90:162-186: 1 = oldpar2, 2 = oldpar1, 3 = newpar
            const double pri2c = GV(sc, R1_PRI2), alpha12 = GV(sc, R1_A12);
            double alpha32, l2;
            if (ny > 1) {
                if (alpha12 == 0.0) alpha32 = 0.0;
                else alpha32 = min1(d_exp(-0.5 * (colsum_diff(ssp1, ssn, s2v, ny, lane) + (pri1 - pri))));
                l2 = -0.5 * (colsum_diff(ssn, ssp2, s2v, ny, lane) + (pri - pri2c));
            } else {
                if (alpha12 == 0.0) alpha32 = 0.0;
                else alpha32 = min1(d_exp(-0.5 * ((GV(ssp1, 0) - GV(ssn, 0)) / L.sigma2 + (pri1 - pri))));
                l2 = -0.5 * ((GV(ssn, 0) - GV(ssp2, 0)) / L.sigma2 + (pri - pri2c));
            }
            const double *iCt = E.iC + (size_t)tile * E.P * 64;
            double *Xq = E.dr_lds ? X : zs_t, *Yq = E.dr_lds ? Y : zs_t + (size_t)d * 64;       // npar > 160: the normal vectors' scratch
            for (int k = 0; k < d; ++k) GV(Xq, k) = GV(new_t, k) - GV(old1_t, k);
            double qa = quadform_sym(iCt, lane, d, Xq, Yq);
            for (int k = 0; k < d; ++k) GV(Xq, k) = GV(cur_t, k) - GV(old1_t, k);
            double qb = quadform_sym(iCt, lane, d, Xq, Yq);
            double q1 = -0.5 * (qa - qb);
            alpha = min1(d_exp(l2 + q1) * (1.0 - alpha32) / (1.0 - alpha12));
        } else {
            alpha = ny > 1 ? d_alpha_cols(ssp1, pri1, ssn, pri, s2v, ny, lane) : d_alpha(GV(ssp1, 0), pri1, GV(ssn, 0), pri, L.sigma2);
        }
        bool reject = true;                               // MCMC_reject, MCMC_DRAM.F90:140-155
        if (alpha >= 1.0) reject = false;
        else if (alpha > 0.0) { double u = rng_uniform(L.g); if (u <= alpha) reject = false; }
        GV(sc, R1_ALPHA) = alpha; GV(sc, R1_REJECT) = reject ? 1.0 : 0.0;
    } else if (MODE == 1 || MODE == 2) {
        gen_normals(L.g, zs_t, lane, d, true);
        // matmulx(R,z)
        if (E.usesvd) gemvN_panels((MODE == 2 ? E.R2f : E.Rf) + (size_t)tile * d * d * 64, zs_t, new_t, cur_t, lane, d, true);
        else trmv_panels<false>((MODE == 2 ? E.R2 : E.R) + (size_t)tile * E.P * 64, zs_t, new_t, cur_t, lane, d, true);
    } else {
        double u = rng_uniform(L.g);
        double s1 = GV(ssp1, 0) / L.sigma2;
        if (ny > 1) { s1 = 0.0; for (int j = 0; j < ny; ++j) s1 = s1 + GV(ssp1, j) / GV(s2v, j); }           // sum(ss1/sigma2)
        GV(sc, R1_CRIT) = -2.0 * d_log(u) + s1 + GV(sc, R1_PRI1);
    }
    lane_store(E, tile, lane, L);
}

} // namespace mcx
