/*
 * oracle/ref/rng_interpose.c -- TEST INFRASTRUCTURE. Not part of the product path.
 *
 * Link-time replacement for the flang runtime's RANDOM_NUMBER / RANDOM_SEED entry
 * points, so that the UNMODIFIED Fortran reference (compiled from /root/reference
 * by oracle/Makefile into oracle/_ref/) draws its uniforms from the pinned
 * Philox stream of oracle/mcx_rng.h instead of the compiler's own generator.
 *
 * Why this is legitimate: the reference defines its stream as "whatever
 * random_number() of the compiler returns" (mcmcrand.F90:55,177), so no two
 * builds of the reference agree on a seed anyway; fixing the stream is the only
 * way to compare accept/reject sequences between the Fortran chain and a GPU lane.
 * Nothing else of the runtime is replaced.
 *
 * Harvest shapes used by the reference: rank-0 real(8) (mcmcrand.F90:104,138,156;
 * MCMC_DRAM.F90:132,151) and rank-1 real(8) (mcmcrand.F90:55,177).  Elements are
 * filled in array-element order, one uniform each.
 *
 * Stream key: environment MCX_SEED (default 0x6D636D63) and MCX_CHAIN (default 0).
 * MCX_RNG_LOG=<file> appends the number of uniforms drawn at exit (debug aid).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ISO_Fortran_binding.h>
#include "../mcx_rng.h"

static mcxo_rng g_rng;
static int g_inited = 0;

static void finish(void)
{
    const char *f = getenv("MCX_RNG_LOG");
    if (f && *f) {
        FILE *fp = fopen(f, "a");
        if (fp) { fprintf(fp, "%llu\n", (unsigned long long)g_rng.n); fclose(fp); }
    }
}

static void ensure_init(void)
{
    if (g_inited) return;
    const char *s = getenv("MCX_SEED");
    const char *c = getenv("MCX_CHAIN");
    uint32_t seed = s ? (uint32_t)strtoul(s, NULL, 0) : MCX_DEFAULT_SEED;
    uint32_t chain = c ? (uint32_t)strtoul(c, NULL, 0) : 0u;
    mcxo_rng_init(&g_rng, seed, chain);
    g_inited = 1;
    atexit(finish);
}

static void fill(const CFI_cdesc_t *d)
{
    ensure_init();
    if (d->elem_len != 8) {
        fprintf(stderr, "rng_interpose: only real(8) harvests are supported (elem_len=%zu)\n", d->elem_len);
        abort();
    }
    if (d->rank == 0) {
        *(double *)d->base_addr = mcxo_uniform(&g_rng);
    } else if (d->rank == 1) {
        char *p = (char *)d->base_addr;
        for (CFI_index_t i = 0; i < d->dim[0].extent; ++i)
            *(double *)(p + i * d->dim[0].sm) = mcxo_uniform(&g_rng);
    } else {
        fprintf(stderr, "rng_interpose: rank %d harvest not supported\n", (int)d->rank);
        abort();
    }
}

void _FortranARandomNumber(const CFI_cdesc_t *harvest, const char *source, int line)
{
    (void)source; (void)line;
    fill(harvest);
}

/* RANDOM_SEED: the stream is keyed by the environment, so PUT is a no-op, SIZE is 1
 * and GET returns the seed word.  (mcmcrand.F90:213-233, 324-327) */
static void put_int(const CFI_cdesc_t *d, long v)
{
    if (!d || !d->base_addr) return;
    if (d->elem_len == 4) *(int32_t *)d->base_addr = (int32_t)v;
    else if (d->elem_len == 8) *(int64_t *)d->base_addr = (int64_t)v;
}
void _FortranARandomSeedSize(const CFI_cdesc_t *size, const char *source, int line)
{ (void)source; (void)line; put_int(size, 1); }
void _FortranARandomSeedPut(const CFI_cdesc_t *put, const char *source, int line)
{ (void)put; (void)source; (void)line; ensure_init(); }
void _FortranARandomSeedGet(const CFI_cdesc_t *get, const char *source, int line)
{ (void)source; (void)line; ensure_init(); put_int(get, (long)(int32_t)g_rng.key[0]); }
void _FortranARandomSeedDefaultPut(void) { ensure_init(); }
void _FortranARandomSeed(const CFI_cdesc_t *size, const CFI_cdesc_t *put, const CFI_cdesc_t *get,
                         const char *source, int line)
{
    if (size && size->base_addr) _FortranARandomSeedSize(size, source, line);
    else if (get && get->base_addr) _FortranARandomSeedGet(get, source, line);
    else (void)put, ensure_init();
}
void _FortranARandomInit(_Bool repeatable, _Bool image_distinct)
{ (void)repeatable; (void)image_distinct; ensure_init(); }
