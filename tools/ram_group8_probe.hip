// Scratch probe (round 5): tools/ram_group_probe.hip with EIGHT lanes per chain -- eight chains per wave, lane l8 of a
// chain owns the columns l8, l8 + 8, ... of the factor (npar 50: seven slots, 218 doubles per lane across VGPRs + AGPRs, one wave per SIMD).
// DCHUD's fifty serial drotg (dchud.f:122-139) then serve EIGHT chains per wave instead of four.  The diagonal element and the work-vector
// element of step i sit in lane i % 8 of each chain; a DPP row holds two chains, so the broadcast is row_newbcast:(i % 8) and
// row_newbcast:(8 + i % 8) and a select on the lane's half of the row.  Times the update alone and checks it against a per-lane loop.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/ram_group8_probe tools/ram_group8_probe.hip && /tmp/ram_group8_probe
#include "../mcmcf90_amd/csrc/mcx_group.hpp"
#include <cstdio>
#include <cstring>
#include <vector>
using namespace mcx;

template <int D>
struct G8 {
    static constexpr int NS = (D + 7) / 8;
    static constexpr int rows(int s) { return D < 8 * (s + 1) ? D : 8 * (s + 1); }
    static constexpr int off(int s) { int o = 0; for (int q = 0; q < s; ++q) o += rows(q); return o; }
    static constexpr int NR = off(NS);
};
template <int N>
MCX_DEV double half_bcast(double x, bool upper) { const double a = row_bcast<N>(x), b = row_bcast<N + 8>(x); return upper ? b : a; }

template <int D>
__global__ __launch_bounds__(64, 1) void chud_group8(const double *Rin, const double *xin, double *Rout, int iters, int d)
{
    using G = G8<D>;
    constexpr int NS = G::NS;
    const int lane = threadIdx.x, l8 = lane & 7, ch8 = lane >> 3;
    const bool upper = (lane & 8) != 0;
    const size_t chain = (size_t)blockIdx.x * 8 + ch8;
    const double *Rc = Rin + chain * (size_t)d * d;          // column-major d x d per chain (upper triangle used)
    double Rr[G::NR], xw[NS], x0[NS];
    sfor<0, NS>([&](auto S) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value;
        const int c = 8 * s + l8;
#pragma unroll
        for (int i = 0; i < G::rows(s); ++i) Rr[G::off(s) + i] = (c < d && i <= c) ? Rc[(size_t)c * d + i] : 0.0;
        x0[s] = (c < d) ? xin[chain * d + c] : 0.0;
    });
    for (int it = 0; it < iters; ++it) {
        sfor<0, NS>([&](auto S) __attribute__((always_inline)) { constexpr int s = decltype(S)::value; xw[s] = x0[s] * (1.0 / (double)(it + 1)); });
        sfor<0, D>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value, si = i / 8, li = i % 8;
            const double a = half_bcast<li>(Rr[G::off(si) + i], upper), b = half_bcast<li>(xw[si], upper);
            double r, c, s;
            d_rotg(a, b, r, c, s);
            {
                const double rij = Rr[G::off(si) + i];
                const double t = c * rij + s * xw[si], xn = c * xw[si] - s * rij;
                Rr[G::off(si) + i] = (l8 == li) ? r : ((l8 > li) ? t : rij);
                xw[si] = (l8 > li) ? xn : xw[si];
            }
            sfor<si + 1, NS>([&](auto S) __attribute__((always_inline)) {
                constexpr int s2 = decltype(S)::value;
                const double rij = Rr[G::off(s2) + i];
                const double t = c * rij + s * xw[s2];
                xw[s2] = c * xw[s2] - s * rij;
                Rr[G::off(s2) + i] = t;
            });
        });
    }
    double *Ro = Rout + chain * (size_t)d * d;
    sfor<0, NS>([&](auto S) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value;
        const int c = 8 * s + l8;
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

template <int D>
static void run()
{
    const int d = D, waves = 1024, nch = waves * 8;
    std::vector<double> R((size_t)nch * d * d, 0.0), x((size_t)nch * d);
    for (int c = 0; c < nch; ++c)
        for (int j = 0; j < d; ++j) { for (int i = 0; i <= j; ++i) R[(size_t)c * d * d + (size_t)j * d + i] = (i == j) ? 1.0 + 0.01 * ((c + j) % 7) : 0.05 * (((i * 7 + j * 3 + c) % 11) - 5) / 5.0; x[(size_t)c * d + j] = 0.3 * (((j * 5 + c) % 13) - 6) / 6.0; }
    double *dR, *dx, *dA, *dB;
    hipMalloc(&dR, R.size() * 8); hipMalloc(&dx, x.size() * 8); hipMalloc(&dA, R.size() * 8); hipMalloc(&dB, R.size() * 8);
    hipMemcpy(dR, R.data(), R.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice);
    hipMemset(dA, 0, R.size() * 8); hipMemset(dB, 0, R.size() * 8);
    hipLaunchKernelGGL(chud_group8<D>, dim3(32), dim3(64), 0, 0, dR, dx, dA, 3, d);
    hipLaunchKernelGGL(chud_lane, dim3(4), dim3(64), 0, 0, dR, dx, dB, 3, d, 256);
    std::vector<double> A((size_t)256 * d * d), B((size_t)256 * d * d);
    hipMemcpy(A.data(), dA, A.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(B.data(), dB, B.size() * 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (int c = 0; c < 256; ++c) for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i) { size_t e = (size_t)c * d * d + (size_t)j * d + i; if (memcmp(&A[e], &B[e], 8)) ++bad; }
    printf("npar %d: eight-lane DCHUD vs per-lane loop, 256 chains x 3 updates: %zu elements differ\n", d, bad);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        const int iters = 100, w = 1024;
        float ms;
        hipEventRecord(e0);
        hipLaunchKernelGGL(chud_group8<D>, dim3(w), dim3(64), 0, 0, dR, dx, dA, iters, d);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("npar %d: DCHUD alone, factor in registers, EIGHT lanes per chain, 1 wave/SIMD: %.3f ms per %d updates of %d chains = %.2f us per wave-update = %.3e chain-updates/s on the chip\n",
                        d, ms, iters, 8 * w, ms * 1e3 / iters, 8.0 * w * iters / (ms * 1e-3));
    }
    hipFree(dR); hipFree(dx); hipFree(dA); hipFree(dB);
}

int main()
{
    run<40>();
    run<48>();
    run<50>();
    return 0;
}
