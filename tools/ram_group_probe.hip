// Scratch probe (round 4): what MCMC_adapt_ram's rank-one update would cost with the factor ON CHIP in the lane-group layout of
// mcx_group.hpp (16 lanes per chain, four chains per wave, R in registers) -- the part of such a kernel that cannot be packed over
// lanes: DCHUD's npar rotations (dchud.f:122-139) are generated one after the other, each from the diagonal element and the work
// vector as the previous rotations left them, and a wave that holds four chains runs the ~64 instructions of a drotg for four useful
// lanes.  Times the update alone (npar = 50, one wave per SIMD: 296 registers of factor) and checks it against a plain per-lane loop.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/ram_group_probe tools/ram_group_probe.hip && /tmp/ram_group_probe
#include "../mcmcf90_amd/csrc/mcx_group.hpp"
#include <cstdio>
#include <cstring>
#include <vector>
using namespace mcx;

template <int D4>
__global__ __launch_bounds__(64, 1) void chud_group(const double *Rin, const double *xin, double *Rout, int iters, int d)
{
    using G = GDims<D4>;
    constexpr int NS = G::NS;
    const int lane = threadIdx.x, l16 = lane & 15, row = lane >> 4;
    const size_t chain = (size_t)blockIdx.x * 4 + row;
    const double *Rc = Rin + chain * (size_t)d * d;          // column-major d x d per chain (upper triangle used)
    double Rr[G::NR], xw[NS], x0[NS];
    sfor<0, NS>([&](auto S) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value;
        const int c = 16 * s + l16;
#pragma unroll
        for (int i = 0; i < G::rows(s); ++i) Rr[G::off(s) + i] = (c < d && i <= c) ? Rc[(size_t)c * d + i] : 0.0;
        x0[s] = (c < d) ? xin[chain * d + c] : 0.0;
    });
    for (int it = 0; it < iters; ++it) {
        sfor<0, NS>([&](auto S) __attribute__((always_inline)) { constexpr int s = decltype(S)::value; xw[s] = x0[s] * (1.0 / (double)(it + 1)); });
        sfor<0, D4>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value, si = i / 16, li = i % 16;
            if (i < d) {
                const double a = row_bcast<li>(Rr[G::off(si) + i]), b = row_bcast<li>(xw[si]);
                double r, c, s;
                d_rotg(a, b, r, c, s);
                {
                    const double rij = Rr[G::off(si) + i];
                    const double t = c * rij + s * xw[si], xn = c * xw[si] - s * rij;
                    Rr[G::off(si) + i] = (l16 == li) ? r : ((l16 > li) ? t : rij);
                    xw[si] = (l16 > li) ? xn : xw[si];
                }
                sfor<si + 1, NS>([&](auto S) __attribute__((always_inline)) {
                    constexpr int s2 = decltype(S)::value;
                    const double rij = Rr[G::off(s2) + i];
                    const double t = c * rij + s * xw[s2];
                    xw[s2] = c * xw[s2] - s * rij;
                    Rr[G::off(s2) + i] = t;
                });
            }
        });
    }
    double *Ro = Rout + chain * (size_t)d * d;
    sfor<0, NS>([&](auto S) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value;
        const int c = 16 * s + l16;
#pragma unroll
        for (int i = 0; i < G::rows(s); ++i) if (c < d && i <= c) Ro[(size_t)c * d + i] = Rr[G::off(s) + i];
    });
}

// the same updates, one lane per chain, factor in global memory (reference for the bits)
__global__ void chud_lane(const double *Rin, const double *xin, double *Rout, int iters, int d, int nchains)
{
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= nchains) return;
    double *R = Rout + (size_t)ch * d * d;
    for (int e = 0; e < d * d; ++e) R[e] = Rin[(size_t)ch * d * d + e];
    double cs[64], sn[64];
    for (int it = 0; it < iters; ++it) {
        for (int j = 0; j < d; ++j) {
            double xj = xin[(size_t)ch * d + j] * (1.0 / (double)(it + 1));
            for (int i = 0; i < j; ++i) { const double rij = R[(size_t)j * d + i]; const double t = cs[i] * rij + sn[i] * xj; xj = cs[i] * xj - sn[i] * rij; R[(size_t)j * d + i] = t; }
            double r; d_rotg(R[(size_t)j * d + j], xj, r, cs[j], sn[j]); R[(size_t)j * d + j] = r;
        }
    }
}

int main()
{
    const int d = 50, waves = 1024, nch = waves * 4;
    std::vector<double> R((size_t)nch * d * d, 0.0), x((size_t)nch * d);
    for (int c = 0; c < nch; ++c) {
        for (int j = 0; j < d; ++j) { for (int i = 0; i <= j; ++i) R[(size_t)c * d * d + (size_t)j * d + i] = (i == j) ? 1.0 + 0.01 * ((c + j) % 7) : 0.05 * (((i * 7 + j * 3 + c) % 11) - 5) / 5.0; x[(size_t)c * d + j] = 0.3 * (((j * 5 + c) % 13) - 6) / 6.0; }
    }
    double *dR, *dx, *dA, *dB;
    hipMalloc(&dR, R.size() * 8); hipMalloc(&dx, x.size() * 8); hipMalloc(&dA, R.size() * 8); hipMalloc(&dB, R.size() * 8);
    hipMemcpy(dR, R.data(), R.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice);
    hipMemset(dA, 0, R.size() * 8); hipMemset(dB, 0, R.size() * 8);
    hipLaunchKernelGGL(chud_group<52>, dim3(64), dim3(64), 0, 0, dR, dx, dA, 3, d);
    hipLaunchKernelGGL(chud_lane, dim3(4), dim3(64), 0, 0, dR, dx, dB, 3, d, 256);
    std::vector<double> A((size_t)256 * d * d), B((size_t)256 * d * d);
    hipMemcpy(A.data(), dA, A.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(B.data(), dB, B.size() * 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (int c = 0; c < 256; ++c) for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i) { size_t e = (size_t)c * d * d + (size_t)j * d + i; if (memcmp(&A[e], &B[e], 8)) ++bad; }
    printf("group-layout DCHUD vs per-lane loop, 256 chains x 3 updates: %zu elements differ\n", bad);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
        for (int w : {1024}) {
            const int iters = 100;
            float ms;
            hipEventRecord(e0);
            hipLaunchKernelGGL(chud_group<52>, dim3(w), dim3(64), 0, 0, dR, dx, dA, iters, d);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            if (rep && w == 1024) printf("DCHUD alone, npar 50, factor in registers, 1 wave/SIMD: %.3f ms per %d updates of %d chains = %.2f us per wave-update = %.3e chain-updates/s on the chip\n",
                                         ms, iters, 4 * w, ms * 1e3 / iters, 4.0 * w * iters / (ms * 1e-3));
        }
    return 0;
}
