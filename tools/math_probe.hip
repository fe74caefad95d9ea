// Scratch probe (round 4): branch-reduced forms of the pinned log / exp (mcx_device.hpp d_log / d_exp, oracle/mcx_math.h).
//   1. bit equality with the branching forms over random bit patterns, dense sweeps of the ranges the sampler uses and edge values;
//   2. polar attempts / s and exp / s of a wave with either form (one and two waves per SIMD).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/math_probe tools/math_probe.hip && /tmp/math_probe
#include "../mcmcf90_amd/csrc/mcx_device.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace mcx;

#ifndef PROBE_OLD
#define PROBE_OLD 0
#endif

__device__ uint64_t splitmix(uint64_t &s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

// mode 0: random 64-bit patterns; 1: x in (0, 1) as the polar method produces them; 2: exp arguments in [-750, 750]; 3: exp arguments near 0
__global__ void check(int mode, uint64_t seed, int per_thread, unsigned long long *bad, double *first_bad)
{
    uint64_t s = seed + 0x1234567ull * (blockIdx.x * blockDim.x + threadIdx.x);
    for (int i = 0; i < per_thread; ++i) {
        uint64_t r = splitmix(s);
        double x;
        if (mode == 0) x = __longlong_as_double((long long)r);
        else if (mode == 1) { double a = 2.0 * ((double)(r >> 11) * 0x1.0p-53) - 1.0; uint64_t r2 = splitmix(s); double b = 2.0 * ((double)(r2 >> 11) * 0x1.0p-53) - 1.0; x = a * a + b * b; if (i & 1) x = x * 0x1.0p-60; }
        else if (mode == 2) x = ((double)(r >> 11) * 0x1.0p-53 - 0.5) * 1500.0;
        else x = ((double)(r >> 11) * 0x1.0p-53 - 0.5) * ((i & 1) ? 3.0 : 1e-6);
        const double l0 = d_log_ref(x), l1 = d_log(x), e0 = d_exp_ref(x), e1 = d_exp(x);
        const bool lb = __double_as_longlong(l0) != __double_as_longlong(l1) && !(l0 != l0 && l1 != l1);
        const bool eb = __double_as_longlong(e0) != __double_as_longlong(e1) && !(e0 != e0 && e1 != e1);
        if (lb || eb) { if (atomicAdd(bad, 1ull) == 0) { first_bad[0] = x; first_bad[1] = lb ? l0 : e0; first_bad[2] = lb ? l1 : e1; first_bad[3] = lb ? 1.0 : 2.0; } }
    }
}

template <int WPS>
__global__ __launch_bounds__(64, WPS) void rate_polar(uint32_t k0, int iters, double *out)
{
    Rng g; g.k0 = k0; g.k1 = blockIdx.x * 64 + threadIdx.x; g.n = 0; g.cblk = 0; g.c2 = g.c3 = 0; g.saved = 0; g.saved_y = 0;
    double acc = 0.0;
    for (int i = 0; i < iters; ++i) { double a, b; if (polar_try(g, a, b)) acc += a + b; }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
template <int WPS>
__global__ __launch_bounds__(64, WPS) void rate_exp(int iters, double *out)
{
    double x = -0.001 * (threadIdx.x + 1), acc = 0.0;
    for (int i = 0; i < iters; ++i) { acc += d_exp(x); x -= 0.37; if (x < -700.0) x += 699.0; }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main()
{
    unsigned long long *bad; double *fb, *out;
    hipMalloc(&bad, 8); hipMalloc(&fb, 32); hipMalloc(&out, 8 * 64 * 4096);
    for (int mode = 0; mode < 4; ++mode) {
        hipMemset(bad, 0, 8);
        hipLaunchKernelGGL(check, dim3(2048), dim3(256), 0, 0, mode, 77ull + mode, 512, bad, fb);
        unsigned long long nb = 0; double h[4];
        hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(h, fb, 32, hipMemcpyDeviceToHost);
        printf("mode %d: %llu values, %llu differ", mode, 2048ull * 256 * 512, nb);
        if (nb) printf(" (first: %s(%a) = %a vs %a)", h[3] == 1.0 ? "log" : "exp", h[0], h[1], h[2]);
        printf("\n");
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
        for (int wps = 1; wps <= 2; ++wps) {
            const int nb = 1024 * wps, iters = 4000;
            float ms;
            hipEventRecord(e0);
            if (wps == 1) hipLaunchKernelGGL(rate_polar<1>, dim3(nb), dim3(64), 0, 0, 5u, iters, out); else hipLaunchKernelGGL(rate_polar<2>, dim3(nb), dim3(64), 0, 0, 5u, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("polar attempts: %d wave(s)/SIMD  %.3f ms  %.3e attempts/s  (%.0f cycles per wave-attempt at 2.1 GHz)\n", wps, ms, (double)nb * 64 * iters / (ms * 1e-3), ms * 1e-3 * 2.1e9 / iters / wps);
            hipEventRecord(e0);
            if (wps == 1) hipLaunchKernelGGL(rate_exp<1>, dim3(nb), dim3(64), 0, 0, iters, out); else hipLaunchKernelGGL(rate_exp<2>, dim3(nb), dim3(64), 0, 0, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("exp:            %d wave(s)/SIMD  %.3f ms  %.3e exp/s       (%.0f cycles per wave-exp)\n", wps, ms, (double)nb * 64 * iters / (ms * 1e-3), ms * 1e-3 * 2.1e9 / iters / wps);
        }
    return 0;
}
