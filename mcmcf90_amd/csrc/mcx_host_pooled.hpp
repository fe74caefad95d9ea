// mcx_host_pooled.hpp -- fetch helpers, signals, the communicator (mcx_comm.hpp), pooled mode's ticks: moments -> all-gather -> tree -> factor (pooled_tick, pooled_ram_tick).
// Part of the ONE translation unit mcx_api.hip (included there, in this order: mcx_host_engine, mcx_host_linalg, mcx_host_launch, mcx_host_adapt, mcx_host_pooled, mcx_host_callbacks); not a stand-alone header.

// ------------------------------------------------------------------ getters
template <typename T>
static int fetch(mcmcx_engine *h, const T *dev, size_t n, std::vector<T> &out)
{
    out.resize(n);
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out.data(), dev, n * sizeof(T), hipMemcpyDeviceToHost));
    return 0;
}
// gather element k of one chain from a tile-interleaved array
template <typename T>
static int fetch_chain_vec(mcmcx_engine *h, const T *dev, int K, int chain, std::vector<T> &out)
{
    int tile = chain / 64, lane = chain % 64;
    std::vector<T> tmp;
    int rc = fetch(h, dev + (size_t)tile * K * 64, (size_t)K * 64, tmp);
    if (rc) return rc;
    out.resize(K);
    for (int k = 0; k < K; ++k) out[k] = tmp[(size_t)k * 64 + lane];
    return 0;
}
static int check_chain(mcmcx_engine *h, int chain)
{
    if (!h) return fail(-1, "null handle");
    if (!h->inited) return fail(-40, "we have not inited");
    if (chain < 0 || chain >= h->cfg.nchains) return fail(-41, "chain index out of range");
    return 0;
}

static void unpack_upper(int d, const std::vector<double> &p, double *colmajor, bool symmetric)
{
    for (int j = 0; j < d; ++j)
        for (int i = 0; i < d; ++i) {
            double v = 0.0;
            if (i <= j) v = p[h_pidx(i, j, d)];
            else if (symmetric) v = p[h_pidx(j, i, d)];
            colmajor[(size_t)i + (size_t)j * d] = v;
        }
}

static volatile sig_atomic_t g_interrupt = 0;
static bool g_sig_installed = false;       // then mcmcx_run waits for each launch, so that a signal is seen at the next boundary
static void on_signal(int) { g_interrupt = 1; }

static int pooled_moments_launch(mcmcx_engine *h, double *dev_dst, int kind = 0, int it = 0);
static int pooled_vec_len(const mcmcx_engine *h, int kind);
#include "mcx_comm.hpp"

// Pooled moments of the chains of ALL ranks, left in h->d_pooled (asynchronous on the engine's stream): local tree ->
// slot `rank` of d_gather -> all-gather over the communicator -> the same pairwise tree over the ranks.  Every rank's
// slot carries one more element behind the vector, the rank's STOP FLAG (1 = a signal was caught here): its sum over the
// ranks comes back with the moments, so the decision to leave a run that has collectives ahead is taken by all ranks at
// the same tick (a rank that returned alone would leave its peers waiting in the next gather).
static int allreduce_moments_enqueue(mcmcx_engine *h, int stage /* 0 all, 1 local part, 2 gather, 3 tree */, int kind = 0, int it = 0,
    double flag = 0.0)
{
    const int len = pooled_vec_len(h, kind), st = len + 1;
    const int nr = h->comm ? h->comm->nranks : 1, rk = h->comm ? h->comm->rank : 0;
    HIPCHK(hipSetDevice(h->cfg.device));
    if (stage == 0 || stage == 1) {
        int rc = pooled_moments_launch(h, h->d_gather + (size_t)rk * st, kind, it); if (rc) return rc;
        h->h_flag = flag;
        hipLaunchKernelGGL(set_double_kernel, dim3(1), dim3(1), 0, h->stream, h->d_gather + (size_t)rk * st + len, flag);
        HIPCHK(hipGetLastError());
    }
    if ((stage == 0 || stage == 2) && h->comm) { int rc = comm_allgather(h->comm, h->d_gather, st, h->stream); if (rc) return rc; }
    if (stage == 0 || stage == 3) {
        hipLaunchKernelGGL(moments_tree_kernel, dim3((st + 255) / 256, 1), dim3(256), 0, h->stream, h->d_gather, nr, st, 1, h->d_pooled);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// several ranks meet in this engine's ticks: its run may only be left at a tick, by agreement (the stop flag above)
static bool collective_run(const mcmcx_engine *h) { return h->pooled && h->comm && h->comm->nranks > 1 && !h->xfn; }

// The pooled statistic vector of `kind` over the chains of ALL ranks, on the host.  With a communicator: local tree ->
// all-gather -> tree over ranks; with the caller's exchange hook (kind 0 only): the hook sums the device buffer.
// When some rank raised its stop flag, h->stop_seen is set: every rank reads the same sum at the same tick, APPLIES that tick
// like any other (all of them hold the same pooled vector) and leaves mcmcx_run behind it -- so a run resumed after
// mcmcx_clear_interrupt continues exactly like one that was never interrupted.
static int pooled_reduce(mcmcx_engine *h, int kind, int it, std::vector<double> &v)
{
    const int len = pooled_vec_len(h, kind);
    v.assign(len + 1, 0.0);
    if (h->xfn && kind != 0) return
        fail(-8, "pooled burn-in scaling and the pooled RAM variant exchange through a communicator (mcmcx_set_comm), not through the mcmcx_set_exchange hook");
    if (!h->xfn) {
        int rc = allreduce_moments_enqueue(h, 0, kind, it, (collective_run(h) && g_interrupt) ? 1.0 : 0.0); if (rc) return rc;
        if ((rc = comm_wait_stream(h->comm, h->stream))) return rc;
        HIPCHK(hipMemcpy(v.data(), h->d_pooled, (size_t)(len + 1) * 8, hipMemcpyDeviceToHost));
    } else {
        double *dst = h->xbuf ? h->xbuf : h->d_moments + (size_t)h->ntiles * len;  // tail of the moments workspace
        int rc = pooled_moments_launch(h, dst, kind, it); if (rc) return rc;
        HIPCHK(hipStreamSynchronize(h->stream));
        h->xfn(h->xuser);                                                            // the caller's own exchange
        HIPCHK(hipMemcpy(v.data(), dst, (size_t)len * 8, hipMemcpyDeviceToHost));
    }
    const double stop = v[len];
    v.resize(len);
    if (collective_run(h) && stop != 0.0) h->stop_seen = true;
    // the vector has been through an exchange: refuse to merge garbage (it would poison the pooled state for the rest of the run)
    if (!(v[0] >= 2.0) || !std::isfinite(v[0])) return fail(-46, "pooled adaptation needs at least 2 chains over all ranks (count = " +
        std::to_string(v[0]) + ")");
    for (int k = 1; k < len;
        ++k) if (!std::isfinite(v[k])) return fail(-46, "pooled adaptation: non-finite pooled statistic at iteration " +
        std::to_string(it));
    return 0;
}

// Delayed rejection in pooled mode (MCMC_adapt.F90:216-225 once for all chains): R2 = R / drscale, iC = dpotri('U', R) on
// the upper triangle of the factor as it stands.  fresh: a new factor (recompute both); otherwise the burn-in scaling
// has already been applied to the tables themselves, as MCMC_adapt.F90:66-78 does.
static int pooled_upload_dr(mcmcx_engine *h, bool fresh)
{
    if (!h->dodr) return 0;
    const int d = h->d, P = h->P;
    if (fresh) {
        std::vector<double> iC(P);
        if (h->usesvd) {
            h->pool_R2 = h->pool_Rf;
            for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i) iC[h_pidx(i, j, d)] = h->pool_Rf[(size_t)j * d + i];
        } else { h->pool_R2 = h->pool_R; iC = h->pool_R; }
        for (auto &v : h->pool_R2) v = v / h->cfg.drscale;
        // the reference stops ("cannot invert cmat"); the old iC stays
        if (host_potri(d, iC) != 0) h->pool_status |= ST_POTRI_FAIL;
        else h->pool_iC = iC;
    }
    std::vector<double> r2 = h->pool_R2;
    if (h->usesvd) r2.resize((size_t)((d + 3) & ~3) * d + PWS, 0.0);
    HIPCHK(hipMemcpyAsync(h->d_sharedR2, r2.data(), r2.size() * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_sharediC, h->pool_iC.data(), (size_t)P * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    // pooled_mfma_kernel<true>: R2 like d_sharedRT (M[s*d + o] = R2(s,o)), iC dense symmetric
    if (h->d_sharedR2T) {
        const int d4 = (d + 3) & ~3;
        std::vector<double> m((size_t)d4 * d + PWS, 0.0), q((size_t)d4 * d + PWS, 0.0);
        // the full factor as it stands: M[s*d + o] = R2f(o, s)
        if (h->usesvd) memcpy(m.data(), h->pool_R2.data(), (size_t)d * d * 8);
        else for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i) m[(size_t)i * d + j] = h->pool_R2[h_pidx(i, j, d)];
        for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i) { const double v = h->pool_iC[h_pidx(i, j, d)]; q[(size_t)i * d + j] = v;
            q[(size_t)j * d + i] = v; }
        HIPCHK(hipMemcpyAsync(h->d_sharedR2T, m.data(), m.size() * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_sharediCd, q.data(), q.size() * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return 0;
}

static int pooled_upload_R(mcmcx_engine *h)
{
    if (h->usesvd) return upload_shared_rf(h);
    HIPCHK(hipMemcpyAsync(h->d_sharedR, h->pool_R.data(), (size_t)h->P * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->d_sharedRT) return upload_shared_rt(h);
    return 0;
}

// chaincmat -> the shared proposal factor (MCMC_calculate_R): dpotf2 + 2.4/sqrt(d), or the pinned SVD for scam; on
// failure the old factor stays (MCMC_adapt.F90:168-171)
static int pooled_factor(mcmcx_engine *h)
{
    const mcmcx_config &c = h->cfg;
    const int d = h->d;
    std::vector<double> cm((size_t)d * d, 0.0), Rp, Cp;
    for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i) cm[(size_t)i + (size_t)j * d] = h->pool_C[h_pidx(i, j, d)];
    if (c.method == MCMCX_METHOD_SCAM) {                // scam_svd of the pooled covariance, MCMC_adapt.F90:189-200
        std::vector<double> U, sd;
        if (host_initial_svd(d, cm, c.condmax, true, U, sd) == 0) { h->pool_U = U; h->pool_std = sd; return upload_shared_u(h); }
        return 0;
    }
    if (h->usesvd) {                                    // covtor_svd of the pooled covariance, MCMC_adapt.F90:203-209
        std::vector<double> Rf, sd, fc;
        if (host_initial_svd(d, cm, c.condmax, false, Rf, sd, &fc) == 0) {
            h->pool_Rf = Rf;
            if (!fc.empty()) for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i) h->pool_C[h_pidx(i, j,
                d)] = fc[(size_t)i + (size_t)j * d];
            int rc = pooled_upload_R(h);
            return rc ? rc : pooled_upload_dr(h, true);
        }
        return 0;
    }
    if (host_initial_R(d, cm, Rp, Cp) == 0) { h->pool_R = Rp; int rc = pooled_upload_R(h); return rc ? rc : pooled_upload_dr(h, true); }
    return 0;
}

// merge the batch of n unit-weight rows (moments v about par0) into (chaincmat, chainmean, chainwsum): covmat's
// weighted update for a whole batch at once, or its batch branch when there is nothing to update (wsum = 0) or when
// `replace` (the AP window: covmat(..., update = .false.), MCMC_adapt.F90:131-133)
static void pooled_merge(mcmcx_engine *h, const std::vector<double> &v, bool replace)
{
    const int d = h->d, P = h->P;
    const double n = v[0];
    std::vector<double> m1(d), mb(d), Cb(P);
    for (int j = 0; j < d; ++j) { m1[j] = v[1 + j] / n; mb[j] = h->par0[j] + m1[j]; }
    for (int j = 0; j < d; ++j)
        for (int i = 0; i <= j; ++i) {
            double s2 = v[1 + d + j * (j + 1) / 2 + i];
            Cb[h_pidx(i, j, d)] = (s2 - n * m1[i] * m1[j]) / (n - 1.0);
        }
    if (replace || !(h->pool_W > 0.0)) {
        h->pool_C = Cb; h->pool_mean = mb; h->pool_W = n;
    } else {
        const double W = h->pool_W, Wn = W + n;
        std::vector<double> dl(d);
        for (int j = 0; j < d; ++j) dl[j] = mb[j] - h->pool_mean[j];
        const double f = W * n / Wn;
        for (int j = 0; j < d; ++j)
            for (int i = 0; i <= j; ++i) {
                const int e = h_pidx(i, j, d);
                h->pool_C[e] = ((W - 1.0) * h->pool_C[e] + (n - 1.0) * Cb[e] + f * dl[i] * dl[j]) / (Wn - 1.0);
            }
        const double g = n / Wn;
        for (int j = 0; j < d; ++j) h->pool_mean[j] = h->pool_mean[j] + g * dl[j];
        h->pool_W = Wn;
    }
}

static void pooled_restart(mcmcx_engine *h)             // chainwsum = initcmatn, chaincmat = cmat0, chainmean = par0
{
    const int d = h->d;
    h->pool_W = (double)h->cfg.initcmatn;
    for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i) h->pool_C[h_pidx(i, j, d)] = h->cmat0[(size_t)i + (size_t)j * d];
    h->pool_mean = h->par0;
}

// Pooled tick: the multi-chain form of MCMC_adapt (MCMC_adapt.F90:60-170).  The N current states of all ranks are a
// batch of N unit-weight rows; `stayed` is summed over the chains.  Every operation below is restated in
// tests/test_gpu_pooled.py.
//   burn-in tick (:60-102): pooled rejection rate sum(stayed) / (N it) against scalelimit -> the shared factor is scaled
//       down / up; in between, greedy restarts from cmat0 and merges the batch, otherwise chaincmat stays, and the
//       factor is recomputed from chaincmat either way (which is what undoes earlier scalings in the reference too)
//   AM tick (:105-159): first time restart from cmat0; merge the batch; AP (adapthist > 1): the batch replaces the
//       covariance instead (the window of the single chain becomes the snapshot of the population)
static int pooled_tick(mcmcx_engine *h, int it, int mode)
{
    const mcmcx_config &c = h->cfg;
    std::vector<double> v;
    if (mode & AD_BURN) {
        int rc = pooled_reduce(h, 1, it, v); if (rc) return rc;
        const double staypc = v[pooled_vec_len(h, 1) - 1] / (v[0] * (double)it);
        const double sf = c.scalefactor;
        if (staypc > 1.0 - c.scalelimit || staypc < c.scalelimit) {
            const bool down = staypc > 1.0 - c.scalelimit;
            for (auto &r : (h->usesvd ? h->pool_Rf : h->pool_R)) r = down ? r / sf : r * sf;
            if (h->dodr) {                                  // R2 and iC are scaled themselves (MCMC_adapt.F90:66-78), not recomputed
                for (auto &r : h->pool_R2) r = down ? r / sf : r * sf;
                for (auto &r : h->pool_iC) r = down ? r * sf * sf : r / sf / sf;
            }
            int rc2 = pooled_upload_R(h);
            return rc2 ? rc2 : pooled_upload_dr(h, false);
        }
        if (c.greedy != 0) { pooled_restart(h); pooled_merge(h, v, false); }
        return pooled_factor(h);
    }
    int rc = pooled_reduce(h, 0, it, v); if (rc) return rc;
    if (it == c.burnintime + c.adaptint + c.adapthist) pooled_restart(h);           // first time: MCMC_adapt.F90:108-114
    pooled_merge(h, v, c.adapthist > 1);
    return pooled_factor(h);
}

// Pooled RAM tick (the multi-chain form of MCMC_adapt_ram, MCMC_run_ram.F90:104-179): every adaptint iterations the
// rank-one statistics of that iteration, one per chain, are averaged over all chains of all ranks and applied to the
// Gram matrix of the shared factor at once:   R'R  <-  R'R + (1/N) sum_c sign(a_c) x_c x_c',
// x_c = u_c / sum(u_c**2) * a_c,  a_c = (alpha_c - alphatarget) / it**nuparam  -- N Cholesky up/downdates of weight 1/N
// folded into one refactorisation.  A Gram matrix that stops being positive definite keeps the old factor (the
// single-chain code stops on a failed downdate; here one bad tick is skipped and flagged).
static int pooled_ram_tick(mcmcx_engine *h, int it)
{
    const int d = h->d, P = h->P;
    std::vector<double> v;
    int rc = pooled_reduce(h, 2, it, v); if (rc) return rc;
    const double n = v[0];
    if (h->usesvd) {
        // condmax > 0: the shared factor is the full matrix Rf of covtor_svd (matutils.F90:378-453), proposals are
        // matmulx(Rf, z) with covariance Rf Rf'.  The same fold on that Gram matrix, refactored the way this factor is made:
        // Rf <- U sqrt(s) of Rf Rf' + (1/N) sum_c sign(a_c) x_c x_c', singular values floored at s_1 / condmax (no 2.4/sqrt(d):
        // the Gram matrix carries the scale already, like the Cholesky form below)
        std::vector<double> S((size_t)d * d, 0.0), Rf, sd;
        for (int j = 0; j < d; ++j)
            for (int i = 0; i <= j; ++i) {
                double acc = 0.0;
                for (int k = 0; k < d; ++k) acc = std::fma(h->pool_Rf[(size_t)k * d + i], h->pool_Rf[(size_t)k * d + j], acc);
                S[(size_t)i + (size_t)j * d] = acc + v[2 + j * (j + 1) / 2 + i] / n;
            }
        if (host_initial_svd(d, S, h->cfg.condmax, false, Rf, sd, nullptr, false) != 0) { h->pool_status |= ST_CHOL_FAIL; return 0; }
        h->pool_Rf = Rf;
        h->pool_alpha = v[1] / n;
        return pooled_upload_R(h);
    }
    std::vector<double> S(P), A;
    for (int j = 0; j < d; ++j)
        for (int i = 0; i <= j; ++i) {
            double acc = 0.0;
            for (int k = 0; k <= i; ++k) acc = std::fma(h->pool_R[h_pidx(k, i, d)], h->pool_R[h_pidx(k, j, d)], acc);
            S[h_pidx(i, j, d)] = acc + v[2 + j * (j + 1) / 2 + i] / n;
        }
    A = S;
    for (int j = 0; j < d; ++j) {                       // dpotf2('U'), the order of host_initial_R
        double dot = 0.0;
        for (int i = 0; i < j; ++i) dot = std::fma(A[h_pidx(i, j, d)], A[h_pidx(i, j, d)], dot);
        double ajj = A[h_pidx(j, j, d)] - dot;
        if (!(ajj > 0.0)) { h->pool_status |= ST_CHOL_FAIL; return 0; }
        double rj = std::sqrt(ajj);
        A[h_pidx(j, j, d)] = rj;
        double rinv = 1.0 / rj;
        for (int k = j + 1; k < d; ++k) {
            double t = 0.0;
            for (int i = 0; i < j; ++i) t = std::fma(A[h_pidx(i, k, d)], A[h_pidx(i, j, d)], t);
            A[h_pidx(j, k, d)] = (A[h_pidx(j, k, d)] - t) * rinv;
        }
    }
    h->pool_R = A;
    h->pool_alpha = v[1] / n;
    return pooled_upload_R(h);
}
