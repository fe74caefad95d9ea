// mcx_host_callbacks.hpp -- the user's host ssfunction / priorfun / checkbounds between the phase kernels (host_eval, host_iteration); MCMC_run1's exchange vectors.
// Part of the ONE translation unit mcx_api.hip (included there, in this order: mcx_host_engine, mcx_host_linalg, mcx_host_launch, mcx_host_adapt, mcx_host_pooled, mcx_host_callbacks); not a stand-alone header.

// Host-callback evaluation of one candidate vector per chain, in chain order, from the calling thread
// (the reference's callbacks keep SAVEd state and are not thread-safe: testcases/mcmcrun.F90:69-70).
// src: tile-interleaved device vector [T][stride][64]; only chains with want != 0 (hx slot) are evaluated.
// what: 0 = checkbounds, priorfun, ssfunction (MCMC_run.F90:47-56); 1 = checkbounds and priorfun only, 2 = ssfunction_er
// with each chain's threshold (the two halves of an early-rejection iteration, MCMC_run_er.F90:54-76)
static int host_eval(mcmcx_engine *h, const double *dev_src, int stride_k, bool use_stage2_flag, int what = 0)
{
    const int d = h->d, T = h->ntiles;
    if (h->tkind == TGT_EXPCOLS) {                      // device-resident response-column target: no host round trip
        hipLaunchKernelGGL(dev_eval_kernel, dim3(T), dim3(64), 0, h->stream, h->E, dev_src, stride_k, use_stage2_flag ? 1 : 0, what);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (h->tkind == TGT_MODULE) {                       // the user's own device code, loaded from a code object
        mcmcx_target_args a;
        a.src = dev_src; a.hev = h->E.hev; a.hx = h->E.hx; a.userdata = h->d_moddata;
        a.stride_k = stride_k; a.npar = d; a.ny = h->ny; a.nhe = NHE - 1 + h->ny; a.nhx = NHX; a.nchains = h->cfg.nchains;
        a.use_stage2 = use_stage2_flag ? 1 : 0; a.what = what;
        size_t asz = sizeof(a);
        void *cfgv[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
        HIPCHK(hipModuleLaunchKernel(h->mod_fn, (unsigned)T, 1, 1, 64, 1, 1, 0, h->stream, nullptr, cfgv));
        return 0;
    }
    const size_t L = (size_t)T * 64;
    const int ny = h->ny, nhe = NHE - 1 + ny;
    const bool src_mapped = h->host_mapped && (dev_src == h->E.cand || h->cs_mapped);
    const bool mapped = h->host_mapped;                  // flags and results in place
    if ((!src_mapped && h->h_cand.resize(L * stride_k)) || (!mapped && (h->h_ev.resize(L * nhe) || (use_stage2_flag
        && h->h_hx.resize(L * NHX)))))
        return fail(-100, "host callbacks: no page-locked memory for the candidates");
    std::vector<double> ssc(ny, 0.0);
    if (!src_mapped) HIPCHK(hipMemcpyAsync(h->h_cand.data(), dev_src, L * stride_k * 8, hipMemcpyDeviceToHost, h->stream));
    if (use_stage2_flag && !mapped) HIPCHK(hipMemcpyAsync(h->h_hx.data(), h->E.hx, h->h_hx.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));             // (also: the previous stage's results have left h_ev)
    const double *h_cand = src_mapped ? dev_src : h->h_cand.data();
    const double *hx = mapped ? h->E.hx : h->h_hx.data();
    double *h_ev = mapped ? h->E.hev : h->h_ev.data();
    memset(h_ev, 0, L * nhe * sizeof(double));
    std::vector<double> th(d);
    if (h->h_ss_batch && !(what == 2 && h->h_ss_er)) {
        // Batched form (opt-in): bounds and prior per chain on this thread, in chain order; then ONE call of the user's
        // ssfunction_batch per worker thread over the chains that need the sum of squares.
        h->h_bidx.clear(); h->h_bth.clear();
        for (int c = 0; c < h->cfg.nchains; ++c) {
            const int t = c / 64, l = c % 64;
            if (use_stage2_flag && hx[((size_t)t * NHX + HX_STAGE2) * 64 + l] == 0.0) continue;
            for (int k = 0; k < d; ++k) th[k] = h_cand[((size_t)t * stride_k + k) * 64 + l];
            int inb = 1; double pri = 0.0;
            if (what != 2) {
                inb = h->h_cb ? h->h_cb(th.data(), d, h->h_user) : 1;
                if (inb) pri = h->h_pri ? h->h_pri(th.data(), d, h->h_user) : 0.0;
            }
            h_ev[((size_t)t * nhe + HE_INB) * 64 + l] = inb ? 1.0 : 0.0;
            h_ev[((size_t)t * nhe + HE_PRI) * 64 + l] = pri;
            if ((what == 0 && inb) || what == 2) { h->h_bidx.push_back(c); h->h_bth.insert(h->h_bth.end(), th.begin(), th.end()); }
        }
        const int n = (int)h->h_bidx.size();
        h->h_bss.assign((size_t)n * ny, 0.0);
        // the first evaluation (MCMC_init's starting point) stays on the calling thread: user code commonly loads its
        // data on first call (testcases/mcmcrun.F90:69-70) -- after that concurrent calls only read it
        const int nt = h->inited ? std::max(1, std::min(h->h_threads, n)) : 1;
        if (nt <= 1) { if (n > 0) h->h_ss_batch(h->h_bth.data(), d, n, ny, h->h_bss.data(), h->h_user); }
        else {
            std::vector<std::thread> pool;
            for (int w = 0; w < nt; ++w) {
                const int lo = (int)((long long)n * w / nt), hi = (int)((long long)n * (w + 1) / nt);
                if (hi > lo) pool.emplace_back([=]() { h->h_ss_batch(h->h_bth.data() + (size_t)lo * d, d, hi - lo, ny,
                    h->h_bss.data() + (size_t)lo * ny, h->h_user); });
            }
            for (auto &t : pool) t.join();
        }
        for (int i = 0; i < n; ++i) {
            const int c = h->h_bidx[i], t = c / 64, l = c % 64;
            for (int j = 0; j < ny; ++j) h_ev[((size_t)t * nhe + HE_SS + j) * 64 + l] = h->h_bss[(size_t)i * ny + j];
        }
        if (!mapped) HIPCHK(hipMemcpyAsync(h->E.hev, h->h_ev.data(), h->h_ev.size() * 8, hipMemcpyHostToDevice, h->stream));
        return 0;
    }
    for (int c = 0; c < h->cfg.nchains; ++c) {
        const int t = c / 64, l = c % 64;
        if (use_stage2_flag && hx[((size_t)t * NHX + HX_STAGE2) * 64 + l] == 0.0) continue;
        for (int k = 0; k < d; ++k) th[k] = h_cand[((size_t)t * stride_k + k) * 64 + l];
        int inb = 1;
        double pri = 0.0;
        std::fill(ssc.begin(), ssc.end(), 0.0);
        if (what == 2) {                                                             // MCMC_ssfunction_er(newpar, sscrit)
            const double crit = hx[((size_t)t * NHX + HX_CRIT) * 64 + l];
            if (h->h_ss_er) h->h_ss_er(th.data(), d, ny, crit, ssc.data(), h->h_user);
            else h->h_ss(th.data(), d, ny, ssc.data(), h->h_user);                   // ssfunction_er0.f90: no er for ss
        } else {
            inb = h->h_cb ? h->h_cb(th.data(), d, h->h_user) : 1;                    // checkbounds0.f90: .true.
            if (inb) {                                                               // MCMC_run.F90:54-56: prior first
                pri = h->h_pri ? h->h_pri(th.data(), d, h->h_user) : 0.0;
                if (what == 0) h->h_ss(th.data(), d, ny, ssc.data(), h->h_user);
            }
        }
        h_ev[((size_t)t * nhe + HE_INB) * 64 + l] = inb ? 1.0 : 0.0;
        h_ev[((size_t)t * nhe + HE_PRI) * 64 + l] = pri;
        for (int j = 0; j < ny; ++j) h_ev[((size_t)t * nhe + HE_SS + j) * 64 + l] = ssc[j];
    }
    if (!mapped) HIPCHK(hipMemcpyAsync(h->E.hev, h->h_ev.data(), h->h_ev.size() * 8, hipMemcpyHostToDevice, h->stream));
    return 0;
}

// fuse_next: iteration it + 1 follows without a tick in between -- its proposal (phase 0; SCAM: component 0's phase 5) rides in this
// iteration's last launch, and h->p0_done tells the next call so (MCMCX_HOST_FUSE=0: one launch per phase, the A/B form the tests compare
// with)
static int host_iteration(mcmcx_engine *h, int it, bool fuse_next)
{
    const dim3 g(h->ntiles), b(64);
    const double *rs = h->d_ramscale + it, *rs0 = h->d_ramscale;
    const size_t lds = lds_step(h);
    const bool fuse = h->sw.host_fuse != 0;
    fuse_next = fuse_next && fuse;
    const bool p0_done = h->p0_done;
    h->p0_done = false;
    if (h->cfg.method == MCMCX_METHOD_SCAM) {           // MCMC_run_scam: npar componentwise proposals, each evaluated by the host
        for (int j = 0; j < h->d; ++j) {
            if (!(j == 0 ? p0_done : fuse)) { hipLaunchKernelGGL((host_phase_kernel<5>), g, b, 0, h->stream, h->E, it, rs, j);
                HIPCHK(hipGetLastError()); }
            int rc = host_eval(h, h->E.cand, h->d, false); if (rc) return rc;
            if (!fuse) hipLaunchKernelGGL((host_phase_kernel<6>), g, b, 0, h->stream, h->E, it, rs, j);
            else if (j + 1 < h->d) hipLaunchKernelGGL((host_phase_seq_kernel<6, 5, -1>), g, b, 0, h->stream, h->E, it, j, it, j + 1, 0, 0,
                rs0);
            else if (fuse_next) { hipLaunchKernelGGL((host_phase_seq_kernel<6, 7, 5>), g, b, 0, h->stream, h->E, it, j, it, 0, it + 1, 0,
                rs0); h->p0_done = true; }
            else hipLaunchKernelGGL((host_phase_seq_kernel<6, 7, -1>), g, b, 0, h->stream, h->E, it, j, it, 0, 0, 0, rs0);
            HIPCHK(hipGetLastError());
        }
        if (!fuse) { hipLaunchKernelGGL((host_phase_kernel<7>), g, b, 0, h->stream, h->E, it, rs, 0); HIPCHK(hipGetLastError()); }
        return 0;
    }
    if (!p0_done) { hipLaunchKernelGGL((host_phase_kernel<0>), g, b, 0, h->stream, h->E, it, rs, 0); HIPCHK(hipGetLastError()); }
    if (h->cfg.method == MCMCX_METHOD_ER) {             // MCMC_run_er: the threshold is drawn between priorfun and ssfunction_er
        int rc = host_eval(h, h->E.cand, h->d, false, 1); if (rc) return rc;
        hipLaunchKernelGGL((host_phase_kernel<3>), g, b, 0, h->stream, h->E, it, rs, 0);
        HIPCHK(hipGetLastError());
        rc = host_eval(h, h->E.cand, h->d, true, 2); if (rc) return rc;
        if (fuse_next) { hipLaunchKernelGGL((host_phase_seq_kernel<4, 0, -1>), g, b, 0, h->stream, h->E, it, 0, it + 1, 0, 0, 0, rs0);
            h->p0_done = true; }
        else hipLaunchKernelGGL((host_phase_kernel<4>), g, b, 0, h->stream, h->E, it, rs, 0);
        HIPCHK(hipGetLastError());
        return 0;
    }
    int rc = host_eval(h, h->E.cand, h->d, false); if (rc) return rc;
    if (fuse_next && !h->dodr) { hipLaunchKernelGGL((host_phase_seq_kernel<1, 0, -1>), g, b, 0, h->stream, h->E, it, 0, it + 1, 0, 0, 0,
        rs0); h->p0_done = true; }
    else hipLaunchKernelGGL((host_phase_kernel<1>), g, b, 0, h->stream, h->E, it, rs, 0);
    HIPCHK(hipGetLastError());
    if (h->dodr) {
        rc = host_eval(h, h->E.cs, 2 * h->d, true); if (rc) return rc;
        if (fuse_next) { hipLaunchKernelGGL((host_phase_seq_kernel<2, 0, -1>), g, b, lds, h->stream, h->E, it, 0, it + 1, 0, 0, 0, rs0);
            h->p0_done = true; }
        else hipLaunchKernelGGL((host_phase_kernel<2>), g, b, lds, h->stream, h->E, it, rs, 0);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// ---- MCMC_run1 / MCMC_run1_er: the arithmetic of one invocation (run1_kernel), all chains at once, vectors row-major per chain
static int run1_check(mcmcx_engine *h)
{
    if (!h) return fail(-1, "null handle");
    if (!h->inited) return fail(-40, "we have not inited");                           // MCMC_run1.F90:55
    if (!h->external) return
        fail(-42, "mcmcx_run1_*: needs mcmcx_set_target_external (the caller evaluates ssfunction / priorfun / checkbounds)");
    return 0;
}
static void run1_put(mcmcx_engine *h, int slot0, int K, const double *src /* [nchains][K] or nullptr */)
{
    const int n1 = 3 * h->d + 3 * h->ny + NR1;
    for (int c = 0; c < h->cfg.nchains; ++c) {
        const int t = c / 64, l = c % 64;
        for (int k = 0; k < K; ++k) h->h_r1[((size_t)t * n1 + slot0 + k) * 64 + l] = src ? src[(size_t)c * K + k] : 0.0;
    }
}
static void run1_get(mcmcx_engine *h, int slot0, int K, double *dst)
{
    const int n1 = 3 * h->d + 3 * h->ny + NR1;
    for (int c = 0; c < h->cfg.nchains; ++c) {
        const int t = c / 64, l = c % 64;
        for (int k = 0; k < K; ++k) dst[(size_t)c * K + k] = h->h_r1[((size_t)t * n1 + slot0 + k) * 64 + l];
    }
}
template <int MODE>
static int run1_launch(mcmcx_engine *h, int drstage)
{
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipMemcpyAsync(h->d_r1, h->h_r1.data(), h->h_r1.size() * 8, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL((run1_kernel<MODE>), dim3(h->ntiles), dim3(64), MODE == 0 ? lds_step(h) : 0, h->stream, h->E, h->d_r1, drstage);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h->h_r1.data(), h->d_r1, h->h_r1.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
