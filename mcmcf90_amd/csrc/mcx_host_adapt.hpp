// mcx_host_adapt.hpp -- the adaptation tick's launch sequence (MCMC_adapt.F90:12-230) and its schedule.
// Part of the ONE translation unit mcx_api.hip (included there, in this order: mcx_host_engine, mcx_host_linalg, mcx_host_launch, mcx_host_adapt, mcx_host_pooled, mcx_host_callbacks); not a stand-alone header.

// the adaptation's SVD one workgroup per chain (mcx_svd.hpp) where its rings and row groups are instantiated
static bool svd_blocked(const mcmcx_engine *h)
{
    if (!h->usesvd || h->pooled || h->cfg.method == MCMCX_METHOD_RAM) return false;
    // test switch: the lane SVD (the engine's form below npar 48 and above 256)
    if (h->sw.svd_lane > 0) return false;
    // (its rings and row groups are instantiated up to npar 256)
    return h->d >= 48 && h->d <= 256;
}
static void launch_adapt(mcmcx_engine *h, int it, int mode)
{
    // one d-vector / the Cholesky's diagonal block (18 kB: eight waves per CU up to npar 36)
    const size_t lds = std::max(lds_bytes(h) / 2, (size_t)36 * 64 * sizeof(double));
    hipLaunchKernelGGL(adapt_pre_kernel, dim3(h->ntiles), dim3(64), 0, h->stream, h->E, it, mode);
    // covmat's batch branch in blocks (adapt_covb_*): the AP window at every adaptation; with initcmatn = 0 the first AM adaptation and
    // the greedy restarts.  Which lanes take it is the lanes' own business (ADF_BATCH); a tick that cannot hold any skips the launches.
    const bool ap = (mode & AD_AM) && h->cfg.adapthist > 1;
    // (test switch: covmat_rows, the lane form)
    const int batch_done = (!(h->sw.cov_batch_rows > 0) &&
                            (ap || (h->cfg.initcmatn == 0 && ((mode & AD_FIRST) || ((mode & AD_BURN) && h->cfg.greedy != 0))))) ? 1 : 0;
    if (batch_done) {
        const int n10 = (h->d + TD - 1) / TD, noff = n10 * (n10 - 1) / 2;
        const unsigned g8 = (unsigned)(8 * ((h->ntiles + 7) / 8));
        hipLaunchKernelGGL(adapt_covb_diag_kernel, dim3(g8 * n10), dim3(64), 0, h->stream, h->E, it, n10);
        if (noff > 0) hipLaunchKernelGGL(adapt_covb_off_kernel, dim3(g8 * noff), dim3(64), 0, h->stream, h->E, it, noff);
    }
    if (!((mode & AD_AM) && h->cfg.adapthist > 1)) {                      // the AP window is a batch recompute: no blocked update
        // blocks of ten (triangular on the diagonal)
        const int n10 = (h->d + TD - 1) / TD, noff = n10 * (n10 - 1) / 2;
        const unsigned g8 = (unsigned)(8 * ((h->ntiles + 7) / 8));
        if (MCX_VARIANT_COV(h, g8, n10, noff, it, mode)) {}
        else {
            hipLaunchKernelGGL(adapt_cov_diag_kernel, dim3(g8 * n10), dim3(64), 0, h->stream, h->E, it, mode, n10);
            if (noff > 0) hipLaunchKernelGGL(adapt_cov_off_kernel, dim3(g8 * noff), dim3(64), 0, h->stream, h->E, it, mode, noff);
        }
    }
    if (!h->d_Gc) {
        if (lds > 160 * 1024) {                             // npar > 320: the work vector in global scratch (slower; no limit)
            if (h->usesvd) hipLaunchKernelGGL((adapt_post_kernel<true, true>), dim3(h->ntiles), dim3(64), 0, h->stream, h->E, it, mode, 0,
                (uint8_t *)nullptr, batch_done);
            else hipLaunchKernelGGL((adapt_post_kernel<false, true>), dim3(h->ntiles), dim3(64), 0, h->stream, h->E, it, mode, 0,
                (uint8_t *)nullptr, batch_done);
            return;
        }
        if (h->usesvd) hipLaunchKernelGGL(adapt_post_kernel<true>, dim3(h->ntiles), dim3(64), lds, h->stream, h->E, it, mode, 0,
            (uint8_t *)nullptr, batch_done);
        else if (h->tile_factor) {
            // dpotf2 (+ dtrti2 / dlauu2 with delayed rejection) with the packed matrices of 4 NW neighbouring chains in LDS, read and
            // written once (mcx_group.hpp: tile_factor_kernel); adapt_post_kernel keeps the covariance bookkeeping (phase 3)
            hipLaunchKernelGGL(adapt_post_kernel<false>, dim3(h->ntiles), dim3(64), lds, h->stream, h->E, it, mode, 3, (uint8_t *)nullptr,
                batch_done);
            const int nc = (h->d + 15) / 16, nw = nc <= 2 ? 4 : nc == 3 ? 2 : 1, ch = 4 * nw;
            const size_t tl = (size_t)ch * (h->P | 1) * sizeof(double) + (size_t)ch * sizeof(int);
            const dim3 tg((unsigned)(8 * ((h->ntiles + 7) / 8) * (64 / ch)));
            switch (nc) {
            case 1: hipLaunchKernelGGL((tile_factor_kernel<1, 4>), tg, dim3(256), tl, h->stream, h->E); break;
            case 2: hipLaunchKernelGGL((tile_factor_kernel<2, 4>), tg, dim3(256), tl, h->stream, h->E); break;
            case 3: hipLaunchKernelGGL((tile_factor_kernel<3, 2>), tg, dim3(128), tl, h->stream, h->E); break;
            default: hipLaunchKernelGGL((tile_factor_kernel<4, 1>), tg, dim3(64), tl, h->stream, h->E); break;
            }
        }
        else hipLaunchKernelGGL(adapt_post_kernel<false>, dim3(h->ntiles), dim3(64), lds, h->stream, h->E, it, mode, 0, (uint8_t *)nullptr,
            batch_done);
        return;
    }
    // large npar with an SVD factor: the factorisation runs one workgroup per chain on chain-major copies, one launch
    // pair per Jacobi sweep (the rotation log lives in Gw, which is free between tile2chain and the next tick)
    const size_t DD = (size_t)h->d * h->d;
    const dim3 tg((unsigned)((DD + 63) / 64), (unsigned)h->ntiles), tg1((unsigned)((h->d + 63) / 64), (unsigned)h->ntiles);
    hipLaunchKernelGGL(adapt_post_kernel<true>, dim3(h->ntiles), dim3(64), lds, h->stream, h->E, it, mode, 1, h->d_need, batch_done);
    hipLaunchKernelGGL(tile2chain_kernel, tg, dim3(256), 0, h->stream, h->E.Gw, h->d_Gc, DD, DD, h->d_need);
    hipLaunchKernelGGL(svd_init_kernel, dim3(h->nlanes), dim3(256), 0, h->stream, h->d_Vc, h->d_state, h->d_need, h->nlanes, h->d);
    for (int sweep = 0; sweep < 60; ++sweep) {
        (void)hipMemsetAsync(h->d_anyrot, 0, sizeof(int), h->stream);
        // the sweep: every later column streamed past a block's pair-lanes through an LDS ring (mcx_svd.hpp) -- all 32 lanes of a row group
        // on pairs up to npar 200 (svd_sweep_stream32_kernel), 24 pair-lanes and a wave of loaders above (svd_sweep_stream_kernel)
        if (h->d <= 200) {
            const int RLs = h->d <= 64 ? 8 : h->d <= 128 ? 16 : 25;
            const size_t lss = (size_t)33 * (8 * RLs + 2) * sizeof(double);
            if (MCX_VARIANT_SVD_SWEEP(h, lss)) {}
            else if (h->d <= 64) hipLaunchKernelGGL(svd_sweep_stream32_kernel<8>, dim3(h->nlanes), dim3(256), lss, h->stream, h->d_Gc,
                (mcx_d2 *)h->E.Gw, h->d_state, h->d_anyrot, h->nlanes, h->d);
            else if (h->d <= 128) hipLaunchKernelGGL(svd_sweep_stream32_kernel<16>, dim3(h->nlanes), dim3(256), lss, h->stream, h->d_Gc,
                (mcx_d2 *)h->E.Gw, h->d_state, h->d_anyrot, h->nlanes, h->d);
            else hipLaunchKernelGGL(svd_sweep_stream32_kernel<25>, dim3(h->nlanes), dim3(256), lss, h->stream, h->d_Gc, (mcx_d2 *)h->E.Gw,
                h->d_state, h->d_anyrot, h->nlanes, h->d);
        } else {
            // pair-lanes of the streamed sweep: whole waves of octets, one wave of loaders at least
            constexpr int svd_sb = 24;
            const int RLs = h->d <= 208 ? 26 : 32;
            const size_t lss = (size_t)(svd_sb + 2) * (8 * RLs + 2) * sizeof(double);
            if (h->d <= 208) hipLaunchKernelGGL(svd_sweep_stream_kernel<26>, dim3(h->nlanes), dim3(256), lss, h->stream, h->d_Gc,
                (mcx_d2 *)h->E.Gw, h->d_state, h->d_anyrot, h->nlanes, h->d, svd_sb);
            else hipLaunchKernelGGL(svd_sweep_stream_kernel<32>, dim3(h->nlanes), dim3(256), lss, h->stream, h->d_Gc, (mcx_d2 *)h->E.Gw,
                h->d_state, h->d_anyrot, h->nlanes, h->d, svd_sb);
        }
        {                                                // the log replayed on V: all 32 lanes of a row group on pairs, any npar
            // four one-wave workgroups per chain, a chain's on one XCD
            const unsigned gv = (unsigned)(32 * ((h->nlanes + 7) / 8));
            const int RP = h->d <= 64 ? 4 : h->d <= 128 ? 8 : h->d <= 208 ? 13 : 16;
            const size_t lsv2 = (size_t)33 * (4 * RP + 6) * sizeof(double);
            if (h->d <= 64) hipLaunchKernelGGL(svd_applyv_stream32_kernel<4>, dim3(gv), dim3(64), lsv2, h->stream, h->d_Vc,
                (const mcx_d2 *)h->E.Gw, h->d_state, h->nlanes, h->d);
            else if (h->d <= 128) hipLaunchKernelGGL(svd_applyv_stream32_kernel<8>, dim3(gv), dim3(64), lsv2, h->stream, h->d_Vc,
                (const mcx_d2 *)h->E.Gw, h->d_state, h->nlanes, h->d);
            else if (h->d <= 208) hipLaunchKernelGGL(svd_applyv_stream32_kernel<13>, dim3(gv), dim3(64), lsv2, h->stream, h->d_Vc,
                (const mcx_d2 *)h->E.Gw, h->d_state, h->nlanes, h->d);
            else hipLaunchKernelGGL(svd_applyv_stream32_kernel<16>, dim3(gv), dim3(64), lsv2, h->stream, h->d_Vc, (const mcx_d2 *)h->E.Gw,
                h->d_state, h->nlanes, h->d);
        }
        int any = 0;
        // (reported by the caller's hipGetLastError)
        if (hipMemcpyAsync(&any, h->d_anyrot, sizeof(int), hipMemcpyDeviceToHost,
            h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) break;
        if (!any) break;
    }
    hipLaunchKernelGGL(svd_finish_kernel, dim3(h->nlanes), dim3(256), 0, h->stream, h->d_Gc, h->d_Vc, h->d_svc, h->d_state, h->nlanes,
        h->d);
    hipLaunchKernelGGL(chain2tile_kernel, tg, dim3(256), 0, h->stream, h->d_Gc, h->E.Vw, DD, DD, h->d_need);
    hipLaunchKernelGGL(chain2tile_kernel, tg1, dim3(256), 0, h->stream, h->d_svc, h->E.cs, (size_t)h->d, (size_t)2 * h->d, h->d_need);
    hipLaunchKernelGGL(adapt_post_kernel<true>, dim3(h->ntiles), dim3(64), lds, h->stream, h->E, it, mode, 2, h->d_need, batch_done);
}

// Which branch of MCMC_adapt fires at iteration `it` (0 = none).  MCMC_adapt.F90:42-46, 60-61, 105.
static int adapt_mode(const mcmcx_config &c, int it)
{
    if (c.method == MCMCX_METHOD_RAM) return 0;
    if (c.doadapt == 0 && c.doburnin == 0) return 0;
    if (c.adaptend > 0 && it > c.adaptend) return 0;
    bool m1 = (c.adaptint != 0) && (it % c.adaptint == 0);
    bool m2 = (c.badaptint != 0) && (it % c.badaptint == 0);
    if (!m1 && !m2) return 0;
    if (it < c.burnintime && c.doburnin != 0 && m2) return AD_BURN;
    if (it >= c.burnintime + c.adaptint + c.adapthist && c.doadapt != 0)
        return AD_AM | ((it == c.burnintime + c.adaptint + c.adapthist) ? AD_FIRST : 0);
    return 0;
}

// pooled RAM: every adaptint iterations once the burn-in is over (MCMC_run_ram.F90:123-131), up to adaptend
static bool pooled_ram_due(const mcmcx_engine *h, int it)
{
    const mcmcx_config &c = h->cfg;
    if (!h->pooled || c.method != MCMCX_METHOD_RAM || c.doadapt == 0 || c.adaptint <= 0) return false;
    if (it < c.burnintime && c.doburnin != 0) return false;
    if (c.adaptend > 0 && it > c.adaptend) return false;
    return it % c.adaptint == 0;
}
