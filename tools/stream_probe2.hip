// Follow-up to tools/stream_probe.hip: what limits the read+write sweep of the tile-interleaved factor at 5.2 TB/s when a
// plain copy reaches ~6.3?  Variants of the same sweep (groups of 8 rows, 2 groups in flight):
//   mode 0  in place, non-temporal loads and stores (the engine's form)
//   mode 1  out of place (read R, write R2), non-temporal
//   mode 2  in place, non-temporal loads, plain stores
//   mode 3  out of place, plain loads and stores
//   mode 4  write only (non-temporal)
//   mode 5  write only (plain)
// plus a grid-stride 16-byte copy for the box's own copy rate, and the tile stride P = 1275 (the engine's at d = 50) next to 1280.
// hipcc --offload-arch=gfx950 -O3 tools/stream_probe2.hip -o tools/_build/stream_probe2 && tools/_build/stream_probe2
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(64, 2) void sweep(double *R, double *R2, int P, int its, int tiles, int perm)
{
    constexpr int G = 8, NB = 2;
    const int lane = threadIdx.x;
    int tile = blockIdx.x;
    if (perm == 1) tile = (tile & 7) * (tiles >> 3) + (tile >> 3);      // XCD x owns a contiguous eighth of the tiles
    const double *Rt = R + (size_t)tile * P * 64;
    double *Wt = ((MODE == 1 || MODE == 3) ? R2 : R) + (size_t)tile * P * 64;
    double acc = 0.0;
    for (int it = 0; it < its; ++it) {
        double r[NB][G];
        const int ng = P / G;
        if (MODE < 4) {
#pragma unroll
            for (int u = 0; u < G; ++u) r[0][u] = (MODE == 3) ? Rt[(size_t)u * 64 + lane] : __builtin_nontemporal_load(&Rt[(size_t)u * 64 + lane]);
        }
        for (int g = 0; g < ng; g += NB) {
#pragma unroll
            for (int s = 0; s < NB; ++s) {
                const int gl = g + s + 1;
                if (MODE < 4 && gl < ng) {
#pragma unroll
                    for (int u = 0; u < G; ++u) {
                        const double *p = &Rt[(size_t)(gl * G + u) * 64 + lane];
                        r[(s + 1) % NB][u] = (MODE == 3) ? *p : __builtin_nontemporal_load(p);
                    }
                }
                if (g + s < ng) {
#pragma unroll
                    for (int u = 0; u < G; ++u) {
                        double t;
                        if (MODE < 4) { t = r[s][u] * 1.0000001 + acc * 1e-30; acc += r[s][u]; }
                        else t = (double)(it + u);
                        double *q = &Wt[(size_t)((g + s) * G + u) * 64 + lane];
                        if (MODE == 2 || MODE == 3 || MODE == 5) *q = t; else __builtin_nontemporal_store(t, q);
                    }
                }
            }
        }
    }
    if (acc == 123.456) Wt[0] = acc;
}

__global__ __launch_bounds__(256) void copy16(const double2 *a, double2 *b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

static const char *names[] = {"in place, nt/nt", "out of place, nt/nt", "in place, nt loads + plain stores", "out of place, plain/plain", "write only, nt", "write only, plain"};

template <int MODE>
static void run(double *R, double *R2, int P, int tiles, int its, int perm)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((sweep<MODE>), dim3(tiles), dim3(64), 0, 0, R, R2, P, its, tiles, perm);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)tiles * (P / 8 * 8) * 64 * 8 * its * (MODE < 4 ? 2 : 1);
    printf("P %4d perm %d  %-36s %7.2f ms  %.2f TB/s\n", P, perm, names[MODE], best, bytes / best / 1e9);
}

int main()
{
    const int tiles = 2048, its = 20;
    const size_t n = (size_t)tiles * 1280 * 64;
    double *R, *R2;
    if (hipMalloc(&R, n * 8) != hipSuccess || hipMalloc(&R2, n * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(R, 0, n * 8); hipMemset(R2, 0, n * 8);
    for (int P : {1280, 1272}) {
        for (int perm = 0; perm < 2; ++perm) {
            run<0>(R, R2, P, tiles, its, perm); run<1>(R, R2, P, tiles, its, perm); run<2>(R, R2, P, tiles, its, perm);
            run<3>(R, R2, P, tiles, its, perm); run<4>(R, R2, P, tiles, its, perm); run<5>(R, R2, P, tiles, its, perm);
        }
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 8192, 65536}) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, (const double2 *)R, (double2 *)R2, n / 2);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        printf("copy16 grid %6d: %7.2f ms  %.2f TB/s (read + written)\n", grid, best, (double)n * 16 * 10 / best / 1e9);
    }
    hipDeviceSynchronize();
    return 0;
}
