// Scratch probe for the RAM factor's layout: one read + one write sweep over a per-chain
// packed factor of P doubles, panels of 10 elements, lane = chain, a fraction of the lanes storing.
//   layout 0: element e of a lane at (e*64 + lane)          -- a row segment of the tile is 512 contiguous bytes
//   layout 1: element e at ((e>>3)*64 + lane)*8 + (e&7)     -- a lane's 8 consecutive elements are one 64-byte sector
// hipcc --offload-arch=gfx950 -O3 tools/layout_probe.hip -o tools/_build/layout_probe && tools/_build/layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int LAYOUT>
__global__ __launch_bounds__(64, 2) void sweep(double *R, int P, int its, unsigned long long storemask_seed, int store_pct)
{
    const int lane = threadIdx.x;
    double *Rt = R + (size_t)blockIdx.x * P * 64;
    // a fixed pseudo-random subset of the lanes stores (the "update lanes"); all lanes load
    unsigned h = (unsigned)(blockIdx.x * 64 + lane) * 2654435761u + (unsigned)storemask_seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const bool st = (int)(h % 100u) < store_pct;
    double acc = 0.0;
    for (int it = 0; it < its; ++it) {
        for (int e0 = 0; e0 + 10 <= P; e0 += 10) {
            double r[10];
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const int e = e0 + u;
                const size_t a = LAYOUT == 0 ? ((size_t)e * 64 + lane) : (((size_t)(e >> 3) * 64 + lane) * 8 + (e & 7));
                r[u] = __builtin_nontemporal_load(&Rt[a]);
            }
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const int e = e0 + u;
                const size_t a = LAYOUT == 0 ? ((size_t)e * 64 + lane) : (((size_t)(e >> 3) * 64 + lane) * 8 + (e & 7));
                const double t = r[u] * 1.0000001 + acc * 1e-30;
                acc += r[u];
                if (st) __builtin_nontemporal_store(t, &Rt[a]);
            }
        }
    }
    if (acc == 123.456) Rt[0] = acc;
}

int main()
{
    const int P = 1280, tiles = 2048, its = 20;
    double *R;
    const size_t n = (size_t)tiles * P * 64;
    hipMalloc(&R, n * 8);
    hipMemset(R, 0, n * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int layout = 0; layout < 2; ++layout)
        for (int pct : {100, 78, 22, 0}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (layout == 0) hipLaunchKernelGGL(sweep<0>, dim3(tiles), dim3(64), 0, 0, R, P, its, 12345ull, pct);
                else hipLaunchKernelGGL(sweep<1>, dim3(tiles), dim3(64), 0, 0, R, P, its, 12345ull, pct);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep == 1) {
                    const double bytes_rd = (double)n * 8 * its, bytes_wr = bytes_rd * pct / 100.0;
                    printf("layout %d, %3d %% of the lanes store: %.2f ms, %.2f TB/s of lane bytes (read %.0f GB + written %.0f GB)\n",
                           layout, pct, ms, (bytes_rd + bytes_wr) / ms / 1e9, bytes_rd / 1e9, bytes_wr / 1e9);
                }
            }
        }
    return 0;
}
