// What a read + write sweep of the tile-interleaved factor can reach with more loads in flight (tools/layout_probe2.hip
// issues one group of 8 loads, then its 8 stores): groups of G elements, the loads of the next NB - 1 groups issued
// before the stores of the current one; non-temporal accesses, all lanes.
// hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o tools/_build/stream_probe && tools/_build/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int G, int NB, bool WRITE>
__global__ __launch_bounds__(64, 2) void sweep(double *R, int P, int its)
{
    const int lane = threadIdx.x;
    double *Rt = R + (size_t)blockIdx.x * P * 64;
    double acc = 0.0;
    for (int it = 0; it < its; ++it) {
        double r[NB][G];
        const int ng = P / G;
#pragma unroll
        for (int s = 0; s < NB - 1; ++s)
#pragma unroll
            for (int u = 0; u < G; ++u) r[s][u] = __builtin_nontemporal_load(&Rt[(size_t)(s * G + u) * 64 + lane]);
        for (int g = 0; g < ng; g += NB) {
#pragma unroll
            for (int s = 0; s < NB; ++s) {
                const int gl = g + s + NB - 1;
                if (gl < ng) {
#pragma unroll
                    for (int u = 0; u < G; ++u) r[(s + NB - 1) % NB][u] = __builtin_nontemporal_load(&Rt[(size_t)(gl * G + u) * 64 + lane]);
                }
                if (g + s < ng) {
#pragma unroll
                    for (int u = 0; u < G; ++u) {
                        const double t = r[s][u] * 1.0000001 + acc * 1e-30;
                        acc += r[s][u];
                        if (WRITE) __builtin_nontemporal_store(t, &Rt[(size_t)((g + s) * G + u) * 64 + lane]);
                    }
                }
            }
        }
    }
    if (acc == 123.456) Rt[0] = acc;
}

template <int G, int NB, bool WRITE>
static void run(double *R, int P, int tiles, int its, size_t n)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((sweep<G, NB, WRITE>), dim3(tiles), dim3(64), 0, 0, R, P, its);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)n * 8 * its * (WRITE ? 2 : 1);
    printf("group %2d x %d in flight, %s: %7.2f ms  %.2f TB/s\n", G, NB, WRITE ? "read+write" : "read only ", best, bytes / best / 1e9);
}

int main()
{
    const int P = 1280, tiles = 2048, its = 20;
    double *R;
    const size_t n = (size_t)tiles * P * 64;
    hipMalloc(&R, n * 8);
    hipMemset(R, 0, n * 8);
    run<8, 1, true>(R, P, tiles, its, n);  run<8, 2, true>(R, P, tiles, its, n);  run<8, 3, true>(R, P, tiles, its, n);  run<8, 4, true>(R, P, tiles, its, n);
    run<10, 2, true>(R, P, tiles, its, n); run<10, 3, true>(R, P, tiles, its, n); run<16, 2, true>(R, P, tiles, its, n); run<16, 4, true>(R, P, tiles, its, n);
    run<8, 1, false>(R, P, tiles, its, n); run<8, 4, false>(R, P, tiles, its, n);
    return 0;
}
