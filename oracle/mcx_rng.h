/*
 * oracle/mcx_rng.h -- TEST INFRASTRUCTURE (CPU oracle). Not part of the product path.
 *
 * The uniform stream that stands in for the Fortran runtime's random_number()
 * (reference call sites: mcmcrand.F90:55,104,138,156,177; MCMC_DRAM.F90:132,151).
 * The reference takes its uniforms from whatever compiler runtime it was built
 * with, so "bit-exact under a fixed seed" is only defined once the stream is
 * pinned.  We pin it as:
 *
 *   Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11; Random123 v1.14 constants)
 *   key     = (seed, chain_id)
 *   counter = (blk_lo, blk_hi, 0, 0)           blk = n >> 1
 *   block   -> 4 x u32 (x0..x3) -> 2 uniforms:
 *       n even: u = (((u64)x1 << 32 | x0) >> 11) * 2^-53
 *       n odd : u = (((u64)x3 << 32 | x2) >> 11) * 2^-53
 *   n = number of uniforms drawn so far by this chain (starts at 0).
 *
 * u is in [0,1), 53 random bits, exactly like a Fortran real(8) harvest.
 * The same stream is fed to the real Fortran reference through
 * oracle/ref/rng_interpose.c and consumed per lane by the HIP engine.
 */
#ifndef MCX_ORACLE_RNG_H
#define MCX_ORACLE_RNG_H
#include <stdint.h>

#define MCX_PHILOX_M0 0xD2511F53u
#define MCX_PHILOX_M1 0xCD9E8D57u
#define MCX_PHILOX_W0 0x9E3779B9u
#define MCX_PHILOX_W1 0xBB67AE85u
#define MCX_DEFAULT_SEED 0x6D636D63u /* "mcmc" */

static inline void mcxo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)MCX_PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)MCX_PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += MCX_PHILOX_W0; k1 += MCX_PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

typedef struct {
    uint32_t key[2];   /* (seed, chain_id) */
    uint64_t n;        /* uniforms consumed so far */
    /* polar-method cache of normal_bm (mcmcrand.F90:172-173), survives across calls */
    int      saved;
    double   saved_y;
} mcxo_rng;

static inline void mcxo_rng_init(mcxo_rng *g, uint32_t seed, uint32_t chain_id)
{
    g->key[0] = seed; g->key[1] = chain_id; g->n = 0; g->saved = 0; g->saved_y = 0.0;
}

static inline double mcxo_uniform(mcxo_rng *g)
{
    uint64_t blk = g->n >> 1;
    uint32_t ctr[4] = { (uint32_t)blk, (uint32_t)(blk >> 32), 0u, 0u };
    uint32_t x[4];
    mcxo_philox4x32_10(ctr, g->key, x);
    uint64_t bits = (g->n & 1) ? (((uint64_t)x[3] << 32) | x[2]) : (((uint64_t)x[1] << 32) | x[0]);
    g->n += 1;
    return (double)(bits >> 11) * 0x1.0p-53;
}

#endif
