// Probe for the RAM factor's layout, round 2: read + write sweeps over a per-chain packed
// factor with a fraction of the lanes loading and a fraction storing.
//   layout 0: element e of a lane at (e*64 + lane)                     8-byte accesses, a row segment = 512 contiguous bytes
//   layout 2: element e at ((e>>3)*64 + lane)*8 + (e&7), accessed as four 16-byte vectors per lane: a lane's 8 consecutive
//             elements are one 64-byte sector, so masked lanes leave their sectors alone
//   layout 3: as 2 with 32-byte units ((e>>2)*64 + lane)*4 + (e&3), two 16-byte vectors
// hipcc --offload-arch=gfx950 -O3 tools/layout_probe2.hip -o tools/_build/layout_probe2 && tools/_build/layout_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

__device__ inline bool pick(unsigned id, unsigned seed, int pct)
{
    unsigned h = id * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    return (int)(h % 100u) < pct;
}

template <int LAYOUT, bool NT>
__global__ __launch_bounds__(64, 2) void sweep(double *R, int P, int its, int load_pct, int store_pct)
{
    const int lane = threadIdx.x;
    double *Rt = R + (size_t)blockIdx.x * P * 64;
    const bool ld = pick(blockIdx.x * 64 + lane, 777u, load_pct);
    const bool st = ld && pick(blockIdx.x * 64 + lane, 12345u, store_pct);
    double acc = 0.0;
    for (int it = 0; it < its; ++it) {
        if (LAYOUT == 0) {
            for (int e0 = 0; e0 + 8 <= P; e0 += 8) {
                double r[8];
                if (ld) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) r[u] = NT ? __builtin_nontemporal_load(&Rt[(size_t)(e0 + u) * 64 + lane]) : Rt[(size_t)(e0 + u) * 64 + lane];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const double t = r[u] * 1.0000001 + acc * 1e-30;
                        acc += r[u];
                        if (st) { if (NT) __builtin_nontemporal_store(t, &Rt[(size_t)(e0 + u) * 64 + lane]); else Rt[(size_t)(e0 + u) * 64 + lane] = t; }
                    }
                }
            }
        } else if (LAYOUT == 2) {
            for (int c = 0; c < P / 8; ++c) {
                d2 *p = (d2 *)(Rt + ((size_t)c * 64 + lane) * 8);
                d2 r[4];
                if (ld) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) r[u] = NT ? __builtin_nontemporal_load(&p[u]) : p[u];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        d2 t;
                        t.x = r[u].x * 1.0000001 + acc * 1e-30; acc += r[u].x;
                        t.y = r[u].y * 1.0000001 + acc * 1e-30; acc += r[u].y;
                        if (st) { if (NT) __builtin_nontemporal_store(t, &p[u]); else p[u] = t; }
                    }
                }
            }
        } else {
            for (int c = 0; c < P / 4; ++c) {
                d2 *p = (d2 *)(Rt + ((size_t)c * 64 + lane) * 4);
                d2 r[2];
                if (ld) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) r[u] = NT ? __builtin_nontemporal_load(&p[u]) : p[u];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        d2 t;
                        t.x = r[u].x * 1.0000001 + acc * 1e-30; acc += r[u].x;
                        t.y = r[u].y * 1.0000001 + acc * 1e-30; acc += r[u].y;
                        if (st) { if (NT) __builtin_nontemporal_store(t, &p[u]); else p[u] = t; }
                    }
                }
            }
        }
    }
    if (acc == 123.456) Rt[0] = acc;
}

template <int LAYOUT, bool NT>
static void run(double *R, int P, int tiles, int its, size_t n)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int cases[][2] = {{100, 100}, {100, 22}, {100, 0}, {78, 100}, {22, 100}};
    for (auto &cs : cases) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((sweep<LAYOUT, NT>), dim3(tiles), dim3(64), 0, 0, R, P, its, cs[0], cs[1]);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        const double rd = (double)n * 8 * its * cs[0] / 100.0, wr = rd * cs[1] / 100.0;
        printf("layout %d %s  load %3d %% store %3d %% of those: %8.2f ms  %.2f TB/s of lane bytes (read %.0f GB + written %.0f GB)\n",
               LAYOUT, NT ? "nt   " : "plain", cs[0], cs[1], best, (rd + wr) / best / 1e9, rd / 1e9, wr / 1e9);
    }
}

int main()
{
    const int P = 1280, tiles = 2048, its = 20;
    double *R;
    const size_t n = (size_t)tiles * P * 64;
    hipMalloc(&R, n * 8);
    hipMemset(R, 0, n * 8);
    run<0, true>(R, P, tiles, its, n);
    run<0, false>(R, P, tiles, its, n);
    run<2, true>(R, P, tiles, its, n);
    run<2, false>(R, P, tiles, its, n);
    run<3, true>(R, P, tiles, its, n);
    run<3, false>(R, P, tiles, its, n);
    return 0;
}
