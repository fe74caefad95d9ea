/*
 * oracle/mcx_math.h -- TEST INFRASTRUCTURE (CPU oracle). Not part of the product path.
 *
 * log() and exp() for the oracle.  The Fortran reference calls the C library
 * (log in mcmcrand.F90:159,183 and MCMC_DRAM.F90:133; exp in MCMC_DRAM.F90:115,
 * 178,185), i.e. an un-pinned third-party libm.  Two libms (glibc on the host,
 * ROCm's OCML on the device) do not round identically, so the oracle pins the
 * two functions as the classic FreeBSD/fdlibm 5.3 algorithms (e_log.c, e_exp.c,
 * (C) 1993 Sun Microsystems, "permission to use, copy, modify, and distribute
 * this software is freely granted"), restated below with the polynomial
 * evaluations written as explicit fma() chains.  Everything else is one IEEE-754
 * operation per C operator (build with -ffp-contract=off), so a device that
 * executes the same operation sequence produces the same bits.  Error < 1 ulp;
 * tests/test_oracle_math.py measures it against the host libm.
 *
 * sqrt() and '/' are IEEE correctly-rounded on both sides and need no pinning.
 *
 * u**e (mcmcrand.F90:105, 138: the shape < 1 branch of the gamma sampler, u uniform in [0,1)) is pinned as
 * exp(e * log(u)) on the two functions above: within |e log u| ulp of libm's pow (< 1e-13 relative for the shapes a
 * run can produce), exactly reproducible on the device.
 */
#ifndef MCX_ORACLE_MATH_H
#define MCX_ORACLE_MATH_H
#include <stdint.h>
#include <string.h>
#include <math.h>

static inline uint64_t mcxo_d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double mcxo_u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
static inline int32_t mcxo_hi(double x) { return (int32_t)(mcxo_d2u(x) >> 32); }
static inline uint32_t mcxo_lo(double x) { return (uint32_t)mcxo_d2u(x); }
static inline double mcxo_sethi(double x, int32_t hi)
{ return mcxo_u2d(((uint64_t)(uint32_t)hi << 32) | (mcxo_d2u(x) & 0xffffffffu)); }

static inline double mcxm_log(double x)
{
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
        two54 = 1.80143985094819840000e+16,
        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
        Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
        Lg7 = 1.479819860511658591e-01;
    int32_t hx = mcxo_hi(x), k = 0, i, j;
    uint32_t lx = mcxo_lo(x);
    if (hx < 0x00100000) {                       /* x < 2**-1022 */
        if (((hx & 0x7fffffff) | lx) == 0) return -INFINITY;   /* log(+-0) = -inf */
        if (hx < 0) return NAN;                  /* log(-#) = NaN */
        k -= 54; x *= two54; hx = mcxo_hi(x);    /* subnormal, scale up */
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    i = (hx + 0x95f64) & 0x100000;
    x = mcxo_sethi(x, hx | (i ^ 0x3ff00000));    /* normalize x or x/2 */
    k += (i >> 20);
    double f = x - 1.0, dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) {           /* |f| < 2**-20 */
        if (f == 0.0) {
            if (k == 0) return 0.0;
            return fma(dk, ln2_hi, dk * ln2_lo);
        }
        double R = (f * f) * fma(-0.33333333333333333, f, 0.5);
        if (k == 0) return f - R;
        return fma(dk, ln2_hi, -((R - dk * ln2_lo) - f));
    }
    double s = f / (2.0 + f);
    double z = s * s;
    i = hx - 0x6147a;
    double w = z * z;
    j = 0x6b851 - hx;
    double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
    double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
    i |= j;
    double R = t2 + t1;
    if (i > 0) {
        double hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return fma(dk, ln2_hi, -((hfsq - fma(s, hfsq + R, dk * ln2_lo)) - f));
    } else {
        if (k == 0) return f - s * (f - R);
        return fma(dk, ln2_hi, -(fma(s, f - R, -(dk * ln2_lo)) - f));
    }
}

static inline double mcxm_exp(double x)
{
    static const double o_threshold = 7.09782712893383973096e+02, u_threshold = -7.45133219101941108420e+02,
        ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
        invln2 = 1.44269504088896338700e+00, twom1000 = 9.33263618503218878990e-302,
        P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
        P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    double hi = 0.0, lo = 0.0, c, t, y;
    int32_t k = 0, xsb;
    uint32_t hx = (uint32_t)mcxo_hi(x);
    xsb = (int32_t)((hx >> 31) & 1);
    hx &= 0x7fffffff;
    if (hx >= 0x40862E42) {                      /* |x| >= 709.78... */
        if (hx >= 0x7ff00000) {
            if (((hx & 0xfffff) | mcxo_lo(x)) != 0) return x + x;   /* NaN */
            return (xsb == 0) ? x : 0.0;                             /* exp(+-inf) */
        }
        if (x > o_threshold) return INFINITY;
        if (x < u_threshold) return 0.0;
    }
    if (hx > 0x3fd62e42) {                       /* |x| > 0.5 ln2 */
        if (hx < 0x3FF0A2B2) {                   /* |x| < 1.5 ln2 */
            if (xsb == 0) { hi = x - ln2HI; lo = ln2LO; k = 1; }
            else          { hi = x + ln2HI; lo = -ln2LO; k = -1; }
        } else {
            k = (int32_t)(fma(invln2, x, (xsb == 0) ? 0.5 : -0.5));
            t = (double)k;
            hi = fma(-t, ln2HI, x);              /* t*ln2HI is exact */
            lo = t * ln2LO;
        }
        x = hi - lo;
    } else if (hx < 0x3e300000) {                /* |x| < 2**-28 */
        return 1.0 + x;
    } else {
        k = 0;
    }
    t = x * x;
    c = x - t * fma(t, fma(t, fma(t, fma(t, P5, P4), P3), P2), P1);
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    if (k >= -1021) {
        return mcxo_u2d(mcxo_d2u(y) + ((uint64_t)(int64_t)k << 52));
    } else {
        y = mcxo_u2d(mcxo_d2u(y) + ((uint64_t)(int64_t)(k + 1000) << 52));
        return y * twom1000;
    }
}

static inline double mcxm_powu(double u, double e) { return mcxm_exp(e * mcxm_log(u)); }

#endif
