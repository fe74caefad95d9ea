// tools/mfma4_probe.hip -- which lane holds what in v_mfma_f64_4x4x4_4b_f64, and in which order it adds its four products: the facts a 4-row tail of the
// pooled kernels' products would need (docs/history/r06.md section 8f).  hipcc --offload-arch=gfx950 -O2 -o mfma4_probe tools/mfma4_probe.hip && ./mfma4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
__global__ void onehot(double *out)
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            out[((size_t)la * 64 + lb) * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        }
}
__global__ void rnd(const double *a, const double *b, const double *c, double *d, int n)
{
    const int lane = threadIdx.x;
    for (int t = 0; t < n; ++t) d[t * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t * 64 + lane], b[t * 64 + lane], c[t * 64 + lane], 0, 0, 0);
}
int main()
{
    double *o; hipMalloc(&o, sizeof(double) * 64 * 64 * 64);
    hipLaunchKernelGGL(onehot, dim3(1), dim3(64), 0, 0, o);
    std::vector<double> h(64 * 64 * 64); hipMemcpy(h.data(), o, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
    // contrib[ld] = the (la, lb) pairs whose product lands in lane ld
    std::vector<std::vector<std::pair<int, int>>> contrib(64);
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) for (int ld = 0; ld < 64; ++ld)
        if (h[((size_t)la * 64 + lb) * 64 + ld] != 0.0) contrib[ld].push_back({la, lb});
    for (int ld = 0; ld < 64; ld += (ld < 4 ? 1 : 7)) {
        printf("D lane %2d <-", ld);
        for (auto &p : contrib[ld]) printf("  A%02d*B%02d", p.first, p.second);
        printf("\n");
    }
    // hypothesis: lane = 16 k + x for A and B (x = 4 block + i resp. 4 block + j), D lane = 16 i + 4 block + j ?  print what holds
    int okA = 1, okB = 1;
    for (int ld = 0; ld < 64; ++ld) {
        if (contrib[ld].size() != 4) { printf("lane %d has %zu products\n", ld, contrib[ld].size()); okA = okB = 0; continue; }
        for (int t = 0; t < 4; ++t) { if (contrib[ld][t].first / 16 != contrib[ld][t].second / 16) okA = 0; }
    }
    printf("every product pairs A and B lanes of the same sixteen (k = lane / 16): %s\n", okA ? "yes" : "no");
    // random data: which order reproduces the bits
    const int n = 512;
    std::vector<double> a(n * 64), b(n * 64), c(n * 64), d(n * 64);
    srand(7);
    for (auto *v : {&a, &b, &c}) for (auto &x : *v) x = (rand() / (double)RAND_MAX - 0.5) * exp((rand() % 40 - 20) * 0.3);
    double *da, *db, *dc, *dd;
    hipMalloc(&da, 8 * n * 64); hipMalloc(&db, 8 * n * 64); hipMalloc(&dc, 8 * n * 64); hipMalloc(&dd, 8 * n * 64);
    hipMemcpy(da, a.data(), 8 * n * 64, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 8 * n * 64, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), 8 * n * 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(rnd, dim3(1), dim3(64), 0, 0, da, db, dc, dd, n);
    hipMemcpy(d.data(), dd, 8 * n * 64, hipMemcpyDeviceToHost);
    int perm[4] = {0, 1, 2, 3};
    do {
        long bad = 0;
        for (int t = 0; t < n; ++t) for (int ld = 0; ld < 64; ++ld) {
            auto pr = contrib[ld];
            std::sort(pr.begin(), pr.end());           // ascending A lane = ascending k under the hypothesis above
            double acc = c[t * 64 + ld];
            for (int q = 0; q < 4; ++q) acc = fma(a[t * 64 + pr[perm[q]].first], b[t * 64 + pr[perm[q]].second], acc);
            if (memcmp(&acc, &d[t * 64 + ld], 8)) ++bad;
        }
        printf("sequential fma chain from c in the order %d %d %d %d of ascending A lanes: %ld of %d differ\n", perm[0], perm[1], perm[2], perm[3], bad, n * 64);
    } while (std::next_permutation(perm, perm + 4));
    return 0;
}
