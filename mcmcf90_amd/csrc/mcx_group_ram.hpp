// mcx_group_ram.hpp -- MCMC_run_ram (MCMC_run_ram.F90:45-179) in the lane-group layout: sixteen lanes per chain, four chains per wave, the
// factor R in REGISTERS for the whole launch and MCMC_adapt_ram's rank-one update / downdate (dchud.f:122-139, dchdd.f:141-179) performed
// on it there.  Included by mcx_api.hip after mcx_group.hpp (round 5; VERDICT round 4, "a few-chains RAM path").
//
// Why: with few chains an iteration is latency.  The lane-per-chain RAM kernels stream the packed factor through L2 twice per iteration,
// one dependent row visit after the other: 184 us per iteration at 64 chains and npar 50, against ~8 us for the reference on one host core.
// Here the factor never leaves the registers between two launches (npar 50: 152 doubles per lane, one wave per SIMD) and an iteration is
// ~20-35 us. It is NOT a throughput kernel -- a wave runs DCHUD's fifty serial drotg for four chains where a lane-per-chain wave runs them
// for 64 (profiles/r04_a/ram_group_probe.txt, profiles/r05_a/ram_group8_probe.txt: measured, closed) -- it saturates at the 4096 chains the
// chip holds at one wave per SIMD (1.68e8 chain-iterations/s at npar 50, where the streaming kernels reach 2.3e8 with 131072 chains and
// more), so the engine takes it up to 16384 chains (32768 from npar 17 on: mcx_api.hip, ram_group_wins;
// profiles/r05_b/ram_group_sweep.txt).
//
// Layout as in group_step_kernel: lane l16 of a chain owns the columns l16, l16 + 16, ... of R (rows 0..column in registers, zeros below
// the diagonal), a vector element k sits in lane k mod 16 (slot k / 16) and reaches the others by row_newbcast.
//   proposal   p_c = sum_{i <= c} R(i,c) z_i ascending in i (v_fmac_f64_dpp chains); after a successful downdate the reference order is the
// diagonal term as a plain product first, then rows c - 1 .. 0 (mcxo_trmv_ut_desc) -- the same uniform walk
//              from the last row down with the lane's own diagonal row selected as the chain's start.
// DCHUD      step i: (R(i,i), x_i) -> drotg on every lane of the chain (broadcast inputs, so c and s are uniform); row i of every later
// column
//              and the work vector rotate -- tools/ram_group_probe.hip's loop, masked per chain.
// DCHDD      forward substitution RIGHT-looking: s_j = (x_j - acc_j) / R(j,j) on the owner lane, broadcast, acc_k = fma(R(j,k), s_j, acc_k)
// for the columns k > j -- every column's ddot receives its terms in ascending row order, as dchdd.f:145 takes them; the classic
// dnrm2 recurrence runs along on the broadcast values; then rotations are generated from the LAST element backwards (dchdd.f:158-167,
//              uniform over the chain) and each is applied at once to row k of the columns >= k (column j's own recurrence xx_j starts at
//              its diagonal, dchdd.f:171-179) -- no rotation is ever stored.
// Every operation on every element is the one ram_update / ram_update_full (mcx_kernels.hpp) perform: bit-equal to the lane kernels and the
// oracle (tests/test_gpu_group.py::test_group_ram_kernel_*).
#pragma once

namespace mcx {

// p = fma(r_i, z_i, p) over the rows i = N-1 .. 0 of a block, lane-uniform walk, with the chain's START at the lane's own diagonal row
// (i == dg: p = r_i * z_i, a plain product; rows above it: untouched).  dg: the lane's diagonal row inside this block, or -1 (the whole
// block lies above / below the lane's column start: `live` says whether the chain has started).
template <int N>
MCX_DEV void blk_fmac_desc(double &p, bool &live, double z, const double *r, int dg)
{
    sfor<0, N>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = N - 1 - decltype(I)::value;
        const double zi = row_bcast<i>(z);
        const double prod = r[i] * zi, f = dfma(r[i], zi, p);
        p = (i == dg) ? prod : (live ? f : p);
        live = live || (i == dg);
    });
}

template <int D4, int TK>
__global__ __launch_bounds__(64, 1) void group_ram_kernel(EngineDev E, int it0, int it1, const double *__restrict__ ramscale,
                                                          const double *__restrict__ g_lamT, uint8_t *accb)
{
    using G = GDims<D4, 16>;
    constexpr int NS = G::NS, CPW = 4, GW = 16;
    // the Gaussian target's precision matrix, [j][i] with pitch D4
    constexpr int LQ = (TK == TGT_GAUSS || TK < 0) ? D4 * D4 : 0;
    __shared__ double lds[CPW * G::ZS + LQ];
    const int lane = threadIdx.x, l16 = lane & 15, row = lane >> 4, d = E.d;
    const int chain = blockIdx.x * CPW + row, tile = chain >> 6, cl = chain & 63;
    const size_t nslots = (size_t)E.ntiles * 64;
    double *zrow = lds + row * G::ZS;
    const double *laml = lds + CPW * G::ZS;
    if constexpr (LQ > 0) {
        if (E.tgt.kind == TGT_GAUSS)
            for (int e = lane; e < D4 * D4; e += 64) { const int j = e / D4, i = e % D4;
                lds[CPW * G::ZS + e] = (i < d && j < d) ? g_lamT[(size_t)j * d + i] : 0.0; }
    }
    // ---- the factor into registers (once per launch; written back at its end)
    double Rr[G::NR];
    {
        const double *Rt = E.R + (size_t)tile * E.P * 64;
        sfor<0, NS>([&](auto S) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            const int c = GW * s + l16;
#pragma unroll
            for (int i = 0; i < G::rows(s); ++i) {
                const bool in = (c < d) && (i <= c);
                const double r = Rt[(in ? (size_t)pidx(i, c, d) : 0) * 64 + cl];
                Rr[G::off(s) + i] = in ? r : 0.0;
            }
        });
    }
    MCX_WAVE_LDS_SYNC();
    double th[NS];
    sfor<0, NS>([&](auto S) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value;
        const int c = GW * s + l16;
        th[s] = (c < d) ? TIDX(E.theta, tile, d, (c < d ? c : 0), cl) : 0.0;
    });
    GChain g;
    g.n = TIDX(E.rngn, tile, 1, 0, cl);
    g.saved = (int)TIDX(E.ictr, tile, NICTR, I_SAVED, cl);
    g.saved_y = TIDX(E.scal, tile, NSCAL, S_SAVEDY, cl);
    g.cb = (g.n >> 1) + (1ull << 62); g.cw[0] = g.cw[1] = g.cw[2] = g.cw[3] = 0u;
    const uint32_t k0 = E.k0, k1 = E.chain_id0 + (uint32_t)chain;
    double ss1 = TIDX(E.scal, tile, NSCAL, S_SS1, cl), pri1 = TIDX(E.scal, tile, NSCAL, S_PRI1, cl);
    double sigma2 = TIDX(E.scal, tile, NSCAL, S_SIGMA2, cl), alpha12 = TIDX(E.scal, tile, NSCAL, S_ALPHA12, cl);
    uint32_t stayed = TIDX(E.ictr, tile, NICTR, I_STAYED, cl), bnd = TIDX(E.ictr, tile, NICTR, I_BNDSTAYED, cl);
    uint32_t chainind = TIDX(E.ictr, tile, NICTR, I_CHAININD, cl), curcount = TIDX(E.ictr, tile, NICTR, I_CURCOUNT, cl);
    uint32_t status = TIDX(E.ictr, tile, NICTR, I_STATUS, cl), downs = TIDX(E.ictr, tile, NICTR, I_DOWNS, cl);
    bool pdesc = TIDX(E.ictr, tile, NICTR, I_PDESC, cl) != 0u;
    const bool adapt = E.doadapt != 0;

    for (int it = it0; it <= it1; ++it) {
        // ---- newpar = MCMC_propose(oldpar, R): the iteration's normals, sum(u**2) in element order (MCMC_run_ram.F90:166), R'z
        double z[NS], cand[NS];
        group_normals<D4, GW>(k0, k1, g, zrow, l16, row, d, true, z);
        double su = 0.0;
        sfor<0, NS>([&](auto S) __attribute__((always_inline)) { constexpr int s = decltype(S)::value; gblk_addchain<GW, G::blk(s)>(su,
            z[s] * z[s]); });
        const bool anydesc = __any(pdesc);
        sfor<0, NS>([&](auto S) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            constexpr int NB = (G::rows(s) + GW - 1) / GW;
            double p = 0.0;
            sfor<0, NB>([&](auto TT) __attribute__((always_inline)) {
                constexpr int tb = decltype(TT)::value;
                constexpr int n = (G::rows(s) - GW * tb) < GW ? (G::rows(s) - GW * tb) : GW;
                gblk_fmac<GW, n>(p, z[tb], &Rr[G::off(s) + GW * tb]);
            });
            if (anydesc) {                                  // (the one proposal after a successful downdate: diagonal first, then upwards)
                double q = 0.0; bool live = false;
                const int c = GW * s + l16;
                sfor<0, NB>([&](auto TT) __attribute__((always_inline)) {
                    constexpr int tb = NB - 1 - decltype(TT)::value;
                    constexpr int n = (G::rows(s) - GW * tb) < GW ? (G::rows(s) - GW * tb) : GW;
                    blk_fmac_desc<n>(q, live, z[tb], &Rr[G::off(s) + GW * tb], c - GW * tb);
                });
                p = pdesc ? q : p;
            }
            cand[s] = th[s] + p;
        });
        // ---- bounds, prior, ss; MCMC_alpha, MCMC_reject (MCMC_run_ram.F90:47-63: an out-of-bounds candidate leaves alpha12 as it was)
        const bool inb = group_inbounds<D4, GW>(E.tgt, cand, l16, row, d);
        const double pri = group_prior<D4, GW>(E.tgt, cand, l16, d);
        const double ss = group_ss<D4, TK, GW>(E.tgt, cand, l16, d, laml);
        bool rej = true, take = false;
        if (!inb) bnd += 1;
        else {
            alpha12 = d_alpha(ss1, pri1, ss, pri, sigma2);
            if (alpha12 >= 1.0) rej = false;
            else if (alpha12 > 0.0) take = true;
        }
        if (__any(take)) {
            const double u = group_uniform<GW>(k0, k1, g, take, row);
            if (take && u <= alpha12) rej = false;
        }
        if (rej) { stayed += 1; curcount += 1; }
        else {
            ss1 = ss; pri1 = pri; chainind += 1; curcount = 1;
            sfor<0, NS>([&](auto S) __attribute__((always_inline)) { constexpr int s = decltype(S)::value; th[s] = cand[s]; });
            if (E.hist) {
                double *h = E.hist + ((size_t)tile * E.wcap + (it % E.wcap)) * (size_t)E.hs * 64;
                sfor<0, NS>([&](auto S) __attribute__((always_inline)) { constexpr int s = decltype(S)::value; const int c = GW * s + l16;
                    if (c < d) h[(size_t)c * 64 + cl] = th[s]; });
                if (l16 == 0) h[(size_t)d * 64 + cl] = ss1;
            }
        }
        if (accb && l16 == 0) accb[(size_t)(it - it0) * nslots + chain] = rej ? (uint8_t)0 : (uint8_t)1;
        // MCMC_updatesigma2: the chain's own sampler on the chain's stream, sixteen identical copies
        if (E.updatesigma) {
            Rng q;
            q.k0 = k0; q.k1 = k1; q.n = g.n; q.cblk = 0; q.c2 = 0; q.c3 = 0; q.saved = g.saved; q.saved_y = g.saved_y;
            const double gm = rng_gamma(q, E.gam_shape, 2.0 / (E.N0S02 + ss1));
            sigma2 = 1.0 / gm;
            g.n = q.n; g.saved = q.saved; g.saved_y = q.saved_y;
        }
        if (E.hist && E.record_s2 && l16 == 0) E.s2hist[((size_t)tile * E.wcap + (it % E.wcap)) * 64 + cl] = sigma2;
        // ---- MCMC_adapt_ram (MCMC_run_ram.F90:104-179): every iteration once the burn-in is over
        if (adapt && !(it < E.burnintime && E.doburnin != 0)) {
            const double a = ramscale[it] * (alpha12 - E.alphatarget);
            const bool up = a >= 0.0, dn = !up;
            downs += dn ? 1u : 0u;
            double xw[NS];
            sfor<0, NS>([&](auto S) __attribute__((always_inline)) {
                constexpr int s = decltype(S)::value;
                const double v = z[s] / su * a;
                xw[s] = up ? v : -v;                          // cholupdate(R, x) / choldowndate(R, x) with x = u / sum(u**2) |a|
            });
            if (up) pdesc = false;
            if (__any(up)) {
                // DCHUD, dchud.f:122-139: step i generates rotation i from (R(i,i), x_i) as the earlier rotations left them and turns row i
                sfor<0, D4>([&](auto I) __attribute__((always_inline)) {
                    constexpr int i = decltype(I)::value, si = i / 16, li = i % 16;
                    if (i < d) {
                        const double aa = row_bcast<li>(Rr[G::off(si) + i]), bb = row_bcast<li>(xw[si]);
                        double r, c, s;
                        d_rotg(aa, bb, r, c, s);
                        {
                            const double rij = Rr[G::off(si) + i];
                            const double t = c * rij + s * xw[si], xn = c * xw[si] - s * rij;
                            const double rnew = (l16 == li) ? r : ((l16 > li) ? t : rij);
                            Rr[G::off(si) + i] = up ? rnew : rij;
                            xw[si] = (up && l16 > li) ? xn : xw[si];
                        }
                        sfor<si + 1, NS>([&](auto S) __attribute__((always_inline)) {
                            constexpr int s2 = decltype(S)::value;
                            const double rij = Rr[G::off(s2) + i];
                            const double t = c * rij + s * xw[s2], xn = c * xw[s2] - s * rij;
                            Rr[G::off(s2) + i] = up ? t : rij;
                            xw[s2] = up ? xn : xw[s2];
                        });
                    }
                });
            }
            if (__any(dn)) {
                // DCHDD, dchdd.f:141-179.  Forward substitution R's = x (right-looking), the norm alongside
                double acc[NS], sv[NS];
                sfor<0, NS>([&](auto S) __attribute__((always_inline)) { acc[decltype(S)::value] = 0.0; sv[decltype(S)::value] = 0.0; });
                double scale = 0.0, ssq = 1.0, s0abs = 0.0;
                sfor<0, D4>([&](auto J) __attribute__((always_inline)) {
                    constexpr int j = decltype(J)::value, sj = j / 16, lj = j % 16;
                    if (j < d) {
                        const double mine = (xw[sj] - acc[sj]) / Rr[G::off(sj) + j];       // the owner lane's is s_j
                        const double s_j = row_bcast<lj>(mine);
                        sv[sj] = (l16 == lj) ? s_j : sv[sj];
                        {   // columns k > j of this slot, and every later slot
                            const double f = dfma(Rr[G::off(sj) + j], s_j, acc[sj]);
                            acc[sj] = (l16 > lj) ? f : acc[sj];
                        }
                        sfor<sj + 1, NS>([&](auto S) __attribute__((always_inline)) { constexpr int s2 = decltype(S)::value;
                            acc[s2] = dfma(Rr[G::off(s2) + j], s_j, acc[s2]); });
                        // dnrm2, classic scale / ssq form (dchdd.f:149)
                        if (j == 0) s0abs = fabs(s_j);
                        if (s_j != 0.0) {
                            const double ax = fabs(s_j);
                            if (scale < ax) { const double q = scale / ax; ssq = 1.0 + ssq * (q * q); scale = ax; }
                            else { const double q = ax / scale; ssq = ssq + q * q; }
                        }
                    }
                });
                const double norm = (d == 1) ? s0abs : scale * sqrt(ssq);
                const bool fail = dn && !(norm < 1.0);
                if (fail) status |= ST_RAM_DOWNDATE_FAIL;            // INFO = -1: R untouched (the reference stops here)
                const bool go = dn && !fail;
                if (dn) pdesc = go;
                if (__any(go)) {
                    // rotations from the last element backwards (dchdd.f:158-167), each applied at once to row k of the columns >= k
                    // (:171-179)
                    double alpha = sqrt(1.0 - norm * norm);
                    double xx[NS];
                    sfor<0, NS>([&](auto S) __attribute__((always_inline)) { xx[decltype(S)::value] = 0.0; });
                    sfor<0, D4>([&](auto K) __attribute__((always_inline)) {
                        constexpr int k = D4 - 1 - decltype(K)::value, sk = k / 16, lk = k % 16;
                        if (k < d) {
                            const double s_k = row_bcast<lk>(sv[sk]);
                            const double sc = alpha + fabs(s_k);
                            const double aa = alpha / sc, bb = s_k / sc;
                            const double nn = sqrt(aa * aa + bb * bb);
                            const double ck = aa / nn, sk_ = bb / nn;
                            alpha = sc * nn;
                            {   // this slot: the columns c >= k
                                const double r = Rr[G::off(sk) + k];
                                const double t = ck * xx[sk] + sk_ * r, rn = ck * r - sk_ * xx[sk];
                                const bool on = go && (l16 >= lk);
                                Rr[G::off(sk) + k] = on ? rn : r;
                                xx[sk] = on ? t : xx[sk];
                            }
                            sfor<sk + 1, NS>([&](auto S) __attribute__((always_inline)) {
                                constexpr int s2 = decltype(S)::value;
                                const double r = Rr[G::off(s2) + k];
                                const double t = ck * xx[s2] + sk_ * r, rn = ck * r - sk_ * xx[s2];
                                Rr[G::off(s2) + k] = go ? rn : r;
                                xx[s2] = go ? t : xx[s2];
                            });
                        }
                    });
                }
            }
        }
    }

    // ---- state and factor back
    {
        double *Rt = E.R + (size_t)tile * E.P * 64;
        sfor<0, NS>([&](auto S) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            const int c = GW * s + l16;
#pragma unroll
            for (int i = 0; i < G::rows(s); ++i) if (c < d && i <= c) Rt[(size_t)pidx(i, c, d) * 64 + cl] = Rr[G::off(s) + i];
        });
    }
    sfor<0, NS>([&](auto S) __attribute__((always_inline)) { constexpr int s = decltype(S)::value; const int c = GW * s + l16;
        if (c < d) TIDX(E.theta, tile, d, c, cl) = th[s]; });
    if (l16 == 0) {
        TIDX(E.scal, tile, NSCAL, S_SIGMA2, cl) = sigma2;
        TIDX(E.rngn, tile, 1, 0, cl) = g.n;
        TIDX(E.ictr, tile, NICTR, I_SAVED, cl) = (uint32_t)g.saved;
        TIDX(E.scal, tile, NSCAL, S_SAVEDY, cl) = g.saved_y;
        TIDX(E.scal, tile, NSCAL, S_SS1, cl) = ss1; TIDX(E.scal, tile, NSCAL, S_PRI1, cl) = pri1;
        TIDX(E.scal, tile, NSCAL, S_ALPHA12, cl) = alpha12;
        TIDX(E.ictr, tile, NICTR, I_STAYED, cl) = stayed; TIDX(E.ictr, tile, NICTR, I_BNDSTAYED, cl) = bnd;
        TIDX(E.ictr, tile, NICTR, I_CHAININD, cl) = chainind; TIDX(E.ictr, tile, NICTR, I_CURCOUNT, cl) = curcount;
        TIDX(E.ictr, tile, NICTR, I_STATUS, cl) = status; TIDX(E.ictr, tile, NICTR, I_DOWNS, cl) = downs;
        TIDX(E.ictr, tile, NICTR, I_PDESC, cl) = pdesc ? 1u : 0u;
    }
}

} // namespace mcx
