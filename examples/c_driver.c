/* examples/c_driver.c -- the C ABI from plain C (C99): 256 chains of the banana target, DRAM, then the pooled mean.
 *
 *   gcc -std=c99 -Iinclude examples/c_driver.c -Lmcmcf90_amd -lmcmcx -Wl,-rpath,$PWD/mcmcf90_amd -o c_driver
 */
#include <stdio.h>
#include <stdlib.h>
#include "mcmcx.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ < 0) { fprintf(stderr, "%s: %s\n", #call, mcmcx_last_error()); return 1; } } while (0)

int main(void)
{
    enum { D = 4, NCHAINS = 256 };
    mcmcx_config cfg;
    mcmcx_handle h;
    double par0[D] = {0, 0, 0, 0}, cmat0[D * D] = {0}, sigma2[1] = {1.0};
    int32_t nobs[1] = {1}, counters[8];
    double *theta = malloc(sizeof(double) * NCHAINS * D), *mom;
    int i, len;

    mcmcx_config_defaults(&cfg);
    cfg.npar = D; cfg.nchains = NCHAINS; cfg.nsimu = 2000; cfg.method = MCMCX_METHOD_DRAM;
    cfg.adaptint = 100; cfg.drscale = 2.0; cfg.updatesigma = 0;
    for (i = 0; i < D; ++i) cmat0[i * D + i] = 0.1;
    CHECK(mcmcx_create(&cfg, &h));
    CHECK(mcmcx_set_par0(h, par0, D));
    CHECK(mcmcx_set_cmat0(h, cmat0, D));
    CHECK(mcmcx_set_sigma2nobs(h, sigma2, nobs, 1));
    CHECK(mcmcx_set_target_banana(h, 0.1));
    CHECK(mcmcx_init(h));
    CHECK(mcmcx_run(h, cfg.nsimu));
    CHECK(mcmcx_sync(h));
    CHECK(mcmcx_get_theta(h, theta));
    CHECK(mcmcx_get_counters(h, 0, counters));
    len = mcmcx_pooled_moments_len(h);
    mom = malloc(sizeof(double) * len);
    CHECK(mcmcx_pooled_moments(h, mom));
    printf("simuind %d, chain 0: stayed %d, dr accepted %d of %d tries\n", mcmcx_simuind(h), counters[0], counters[2], counters[3]);
    printf("pooled mean over %g chains:", mom[0]);
    for (i = 0; i < D; ++i) printf(" %.4f", mom[1 + i] / mom[0]);
    printf("\n");
    mcmcx_destroy(h);
    free(theta); free(mom);
    return 0;
}
