// mcx_device.hpp -- per-lane device primitives of the adaptive-Metropolis engine (gfx950).
//
// One lane = one chain.  Everything here is the device statement of a reference
// routine (cited per function); the arithmetic follows the conventions fixed in
// reference Fortran -> one IEEE op per operator (this file is
// compiled with -ffp-contract=off), BLAS-type accumulations -> explicit fma chains,
// log/exp -> the pinned fdlibm-style sequences below, so that a lane reproduces
// the CPU chain bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcx {

#define MCX_DEV __device__ __forceinline__

// ---------------------------------------------------------------- bits
MCX_DEV int32_t hi32(double x) { return __double2hiint(x); }
MCX_DEV uint32_t lo32(double x) { return (uint32_t)__double2loint(x); }
MCX_DEV double set_hi32(double x, int32_t hi) { return __hiloint2double(hi, __double2loint(x)); }
MCX_DEV double add_exp(double y, int32_t k)
{ return __longlong_as_double(__double_as_longlong(y) + ((long long)k << 52)); }
MCX_DEV double dfma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// ---------------------------------------------------------------- log / exp (pinned; see oracle/mcx_math.h)
MCX_DEV double d_log_ref(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
        two54 = 1.80143985094819840000e+16,
        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
        Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
        Lg7 = 1.479819860511658591e-01;
    int32_t hx = hi32(x), k = 0, i, j;
    uint32_t lx = lo32(x);
    if (hx < 0x00100000) {
        if (((hx & 0x7fffffff) | lx) == 0) return -__builtin_inf();
        if (hx < 0) return __builtin_nan("");
        k -= 54; x *= two54; hx = hi32(x);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    i = (hx + 0x95f64) & 0x100000;
    x = set_hi32(x, hx | (i ^ 0x3ff00000));
    k += (i >> 20);
    double f = x - 1.0, dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) {
        if (f == 0.0) {
            if (k == 0) return 0.0;
            return dfma(dk, ln2_hi, dk * ln2_lo);
        }
        double R = (f * f) * dfma(-0.33333333333333333, f, 0.5);
        if (k == 0) return f - R;
        return dfma(dk, ln2_hi, -((R - dk * ln2_lo) - f));
    }
    double s = f / (2.0 + f);
    double z = s * s;
    i = hx - 0x6147a;
    double w = z * z;
    j = 0x6b851 - hx;
    double t1 = w * dfma(w, dfma(w, Lg6, Lg4), Lg2);
    double t2 = z * dfma(w, dfma(w, dfma(w, Lg7, Lg5), Lg3), Lg1);
    i |= j;
    double R = t2 + t1;
    if (i > 0) {
        double hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dfma(dk, ln2_hi, -((hfsq - dfma(s, hfsq + R, dk * ln2_lo)) - f));
    } else {
        if (k == 0) return f - s * (f - R);
        return dfma(dk, ln2_hi, -(dfma(s, f - R, -(dk * ln2_lo)) - f));
    }
}

MCX_DEV double d_exp_ref(double x)
{
    const double o_threshold = 7.09782712893383973096e+02, u_threshold = -7.45133219101941108420e+02,
        ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
        invln2 = 1.44269504088896338700e+00, twom1000 = 9.33263618503218878990e-302,
        P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
        P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    double hi = 0.0, lo = 0.0, c, t, y;
    int32_t k = 0, xsb;
    uint32_t hx = (uint32_t)hi32(x);
    xsb = (int32_t)((hx >> 31) & 1);
    hx &= 0x7fffffff;
    if (hx >= 0x40862E42) {
        if (hx >= 0x7ff00000) {
            if (((hx & 0xfffff) | lo32(x)) != 0) return x + x;
            return (xsb == 0) ? x : 0.0;
        }
        if (x > o_threshold) return __builtin_inf();
        if (x < u_threshold) return 0.0;
    }
    if (hx > 0x3fd62e42) {
        if (hx < 0x3FF0A2B2) {
            if (xsb == 0) { hi = x - ln2HI; lo = ln2LO; k = 1; }
            else          { hi = x + ln2HI; lo = -ln2LO; k = -1; }
        } else {
            k = (int32_t)(dfma(invln2, x, (xsb == 0) ? 0.5 : -0.5));
            t = (double)k;
            hi = dfma(-t, ln2HI, x);
            lo = t * ln2LO;
        }
        x = hi - lo;
    } else if (hx < 0x3e300000) {
        return 1.0 + x;
    } else {
        k = 0;
    }
    t = x * x;
    c = x - t * dfma(t, dfma(t, dfma(t, dfma(t, P5, P4), P3), P2), P1);
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    if (k >= -1021) return add_exp(y, k);
    y = add_exp(y, k + 1000);
    return y * twom1000;
}

// The same two functions as the kernels call them: value for value d_log_ref / d_exp_ref (tools/math_probe.hip compares them over
// random bit patterns and dense sweeps; tests/test_gpu_primitives.py against the oracle), with the case distinctions of the common
// range written as selects.  A wave of 64 chains takes every one of fdlibm's branches at nearly every call -- k = 0 or not, the two
// forms of the final correction, |x| below or above 1.5 ln 2 -- so the branching form runs all of them one after the other behind
// exec masks, each with its scalar bookkeeping; here every lane computes the few operations in which the cases differ and picks.
// Only the rare ranges stay branches (a wave skips them when no lane is there): subnormal / zero / negative / non-finite arguments
// and |x - 1| < 2**-20 for the logarithm, |x| >= 709.78 and non-finite arguments for the exponential.
MCX_DEV double d_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
        Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
        Lg7 = 1.479819860511658591e-01;
#ifdef MCX_MATH_REF
    return d_log_ref(x);                                                 // (A/B: tools/math_probe.hip)
#endif
    int32_t hx = hi32(x);
    if (hx < 0x00100000 || hx >= 0x7ff00000) return d_log_ref(x);       // subnormal, zero, negative, infinite, NaN
    int32_t k = (hx >> 20) - 1023;
    hx &= 0x000fffff;
    int32_t i = (hx + 0x95f64) & 0x100000;
    x = set_hi32(x, hx | (i ^ 0x3ff00000));
    k += (i >> 20);
    const double f = x - 1.0, dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) {                                   // |f| < 2**-20
        if (f == 0.0) {
            if (k == 0) return 0.0;
            return dfma(dk, ln2_hi, dk * ln2_lo);
        }
        const double R = (f * f) * dfma(-0.33333333333333333, f, 0.5);
        if (k == 0) return f - R;
        return dfma(dk, ln2_hi, -((R - dk * ln2_lo) - f));
    }
    const double s = f / (2.0 + f);
    const double z = s * s;
    i = hx - 0x6147a;
    const double w = z * z;
    const int32_t j = 0x6b851 - hx;
    const double t1 = w * dfma(w, dfma(w, Lg6, Lg4), Lg2);
    const double t2 = z * dfma(w, dfma(w, dfma(w, Lg7, Lg5), Lg3), Lg1);
    i |= j;
    const double R = t2 + t1;
    const bool A = i > 0;
    const double hfsq = 0.5 * f * f;
    const double t = A ? hfsq + R : f - R;
    const double c = dk * ln2_lo;
    // k != 0:  A: dk ln2_hi - ((hfsq - (s (hfsq + R) + dk ln2_lo)) - f)      else: dk ln2_hi - ((s (f - R) - dk ln2_lo) - f)
    const double m = dfma(s, t, A ? c : -c);
    const double rk = dfma(dk, ln2_hi, -((A ? hfsq - m : m) - f));
    // k == 0:  A: f - (hfsq - s (hfsq + R))                                    else: f - s (f - R)
    const double m0 = s * t;
    const double r0 = f - (A ? hfsq - m0 : m0);
    return (k == 0) ? r0 : rk;
}

MCX_DEV double d_exp(double x)
{
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
        invln2 = 1.44269504088896338700e+00, twom1000 = 9.33263618503218878990e-302,
        P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
        P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
#ifdef MCX_MATH_REF
    return d_exp_ref(x);
#endif
    uint32_t hx = (uint32_t)hi32(x);
    const bool neg = (hx >> 31) != 0u;
    hx &= 0x7fffffff;
    if (hx >= 0x40862E42) return d_exp_ref(x);                           // |x| >= 709.78, infinite, NaN
    // |x| > 0.5 ln 2: k = +-1 below 1.5 ln 2 (hi = x -+ ln2HI, lo = +-ln2LO), else k = int(x / ln 2 +- 0.5), hi = x - k ln2HI, lo = k ln2LO
    // --
    // the first form is the second with k forced (fma(-(+-1), ln2HI, x) rounds x -+ ln2HI once, (+-1) ln2LO is exact); k = 0 otherwise,
    // for which hi = x, lo = 0 and hi - lo = x
    const bool big = hx > 0x3fd62e42, mid = hx < 0x3FF0A2B2;
    int32_t k = (int32_t)(dfma(invln2, x, neg ? -0.5 : 0.5));
    k = mid ? (neg ? -1 : 1) : k;
    k = big ? k : 0;
    const double t = (double)k;
    const double hi = dfma(-t, ln2HI, x);
    const double lo = t * ln2LO;
    const double xr = hi - lo;
    const double tt = xr * xr;
    const double c = xr - tt * dfma(tt, dfma(tt, dfma(tt, dfma(tt, P5, P4), P3), P2), P1);
    // k == 0: 1 - ((x c) / (c - 2) - x);  else y = 1 - ((lo - (x c) / (2 - c)) - hi): one division, a / (-b) = -(a / b) exactly
    const double q = (xr * c) / (2.0 - c);
    const double r0 = 1.0 - ((-q) - xr);
    const double y = 1.0 - ((lo - q) - hi);
    const bool deep = k < -1021;
    double ys = add_exp(y, deep ? k + 1000 : k);
    ys = deep ? ys * twom1000 : ys;
    double r = (k == 0) ? r0 : ys;
    r = (!big && hx < 0x3e300000) ? 1.0 + x : r;                         // |x| < 2**-28
    return r;
}

// ---------------------------------------------------------------- RNG: Philox4x32-10 stream per lane
// Stands in for the Fortran runtime's random_number (mcmcrand.F90:55,104,138,156,177;
// MCMC_DRAM.F90:132,151).  key = (seed, chain id); uniform #n comes from block n>>1.
struct Rng {
    uint32_t k0, k1;
    uint64_t n;         // uniforms consumed
    uint32_t c2, c3;    // words 2,3 of the block that holds uniform n when n is odd
    uint64_t cblk;      // block index the cached words belong to (+1; 0 = none)
    int saved;          // normal_bm cache flag, mcmcrand.F90:172-173
    double saved_y;
};

MCX_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1,
                           uint32_t &o0, uint32_t &o1, uint32_t &o2, uint32_t &o3)
{
    uint32_t c2 = 0u, c3 = 0u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

MCX_DEV double bits_to_uniform(uint32_t lo, uint32_t hi)
{
    uint64_t b = ((uint64_t)hi << 32) | lo;
    return (double)(b >> 11) * 0x1.0p-53;
}

MCX_DEV double rng_uniform(Rng &g)
{
    uint64_t blk = g.n >> 1;
    double u;
    if ((g.n & 1) && g.cblk == blk + 1) {
        u = bits_to_uniform(g.c2, g.c3);
    } else {
        uint32_t x0, x1, x2, x3;
        philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), g.k0, g.k1, x0, x1, x2, x3);
        if (g.n & 1) u = bits_to_uniform(x2, x3);
        else { u = bits_to_uniform(x0, x1); g.c2 = x2; g.c3 = x3; g.cblk = blk + 1; }
    }
    g.n += 1;
    return u;
}

// two uniforms in stream order (call random_number(x), x(2): mcmcrand.F90:177)
MCX_DEV void rng_uniform2(Rng &g, double &u1, double &u2)
{
    if ((g.n & 1) == 0) {
        uint64_t blk = g.n >> 1;
        uint32_t x0, x1, x2, x3;
        philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), g.k0, g.k1, x0, x1, x2, x3);
        u1 = bits_to_uniform(x0, x1); u2 = bits_to_uniform(x2, x3);
        g.n += 2;
    } else {
        u1 = rng_uniform(g); u2 = rng_uniform(g);
    }
}

// One polar-method attempt (mcmcrand.F90:176-187).  Returns true when the pair is accepted;
// first = z*x(2) is returned by this call of normal_bm, second = z*x(1) by the next one.
MCX_DEV bool polar_try(Rng &g, double &first, double &second)
{
    double x1, x2;
    rng_uniform2(g, x1, x2);
    x1 = 2.0 * x1 - 1.0; x2 = 2.0 * x2 - 1.0;
    double xx = x1 * x1 + x2 * x2;
    bool ok = (xx < 1.0) && (xx != 0.0);
    if (ok) {
        double z = sqrt(-2.0 * d_log(xx) / xx);
        second = z * x1; first = z * x2;
    }
    return ok;
}

// normal_bm() for one lane (used by the gamma sampler, where consumption is data dependent)
MCX_DEV double rng_normal(Rng &g)
{
    if (g.saved) { g.saved = 0; return g.saved_y; }
    double a, b;
    while (!polar_try(g, a, b)) {}
    g.saved_y = b; g.saved = 1;
    return a;
}

// random_gamma / gammar_mt (mcmcrand.F90:86-162).  a < 1 (:102-105): u first, then gammar_mt(1+a, b) * u**(1/a), with
// u**e pinned as exp(e log u) (oracle/mcx_math.h).
MCX_DEV double rng_gamma(Rng &g, double a, double b)
{
    double boost = 1.0;
    const bool small = a < 1.0;
    if (small) { double u0 = rng_uniform(g); boost = d_exp((1.0 / a) * d_log(u0)); a = 1.0 + a; }
    double d = a - 1.0 / 3.0;
    double c = 1.0 / sqrt(9.0 * d);
    double x, v, u;
    for (;;) {
        do { x = rng_normal(g); v = 1.0 + c * x; } while (!(v > 0.0));
        v = (v * v) * v;
        u = rng_uniform(g);
        double x2 = x * x;
        if (u < 1.0 - 0.0331 * (x2 * x2)) break;
        if (d_log(u) < 0.5 * x2 + d * (1.0 - v + d_log(v))) break;
    }
    const double y = b * d * v;
    return small ? y * boost : y;
}

// ---------------------------------------------------------------- small BLAS pieces (pinned netlib forms)
// classic drotg as used by dchud.f:138 (r, c, s only)
MCX_DEV void d_rotg(double a, double b, double &r, double &c, double &s)
{
    double roe = b;
    if (fabs(a) > fabs(b)) roe = a;
    double scale = fabs(a) + fabs(b);
    if (scale == 0.0) { c = 1.0; s = 0.0; r = 0.0; }
    else {
        double t1 = a / scale, t2 = b / scale;
        double rr = scale * sqrt(t1 * t1 + t2 * t2);
        rr = copysign(1.0, roe) * rr;
        c = a / rr; s = b / rr; r = rr;
    }
}

// MCMC_alpha, MCMC_DRAM.F90:100-118 (nycol = 1)
MCX_DEV double d_alpha(double ss1, double pri1, double ss2, double pri2, double sigma2)
{
    double tst = -0.5 * ((ss2 - ss1) / sigma2 + (pri2 - pri1));
    double a;
    if (tst >= 0.0) a = 1.0;
    else if (tst < -708.39641853226408) a = 0.0;
    else a = d_exp(tst);
    return a;
}

MCX_DEV double min1(double x) { return (1.0 < x) ? 1.0 : x; }

} // namespace mcx
