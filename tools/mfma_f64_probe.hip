// Scratch probe: (1) is v_mfma_f64_16x16x4_f64 bit-identical to an ascending-k fma chain?  (2) its rate vs v_fma_f64.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <cstdint>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void probe(const double *A, const double *B, const double *C, double *D, int nrep)
{
    // A[16][4*nrep], B[4*nrep][16], C[16][16]
    const int l = threadIdx.x;
    d4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)];
    for (int s = 0; s < nrep; ++s) {
        double a = A[(l & 15) * (4 * nrep) + 4 * s + (l >> 4)];
        double b = B[(4 * s + (l >> 4)) * 16 + (l & 15)];
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

template <int MODE>   // 0 mfma, 1 vector fma, 2 both (even waves mfma, odd waves fma)
__global__ void rate(double *out, int iters)
{
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    double a = 1.0 + l * 1e-9, b = 1.0 - l * 1e-9;
    double sum = 0;
    bool mf = (MODE == 0) || (MODE == 2 && (w & 1) == 0);
    if (mf) {
        d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
        sum = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        double c[16];
        for (int u = 0; u < 16; ++u) c[u] = u;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int u = 0; u < 16; ++u) c[u] = __builtin_fma(a, b, c[u]);
        }
        for (int u = 0; u < 16; ++u) sum += c[u];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

int main()
{
    const int nrep = 3;
    std::vector<double> A(16 * 4 * nrep), B(4 * nrep * 16), C(256), D(256);
    srand(7);
    auto rnd = []() { double m = (rand() / (double)RAND_MAX) * 2 - 1; int e = rand() % 40 - 20; return std::ldexp(m, e); };
    for (auto &v : A) v = rnd(); for (auto &v : B) v = rnd(); for (auto &v : C) v = rnd();
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, 2048); hipMalloc(&dD, 2048);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, nrep);
    hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
    int asc = 0, desc = 0, mul = 0, pair = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        const int K = 4 * nrep;
        double x = C[i * 16 + j], y = C[i * 16 + j], z = C[i * 16 + j], p = C[i * 16 + j];
        for (int k = 0; k < K; ++k) x = std::fma(A[i * K + k], B[k * 16 + j], x);
        for (int s = 0; s < nrep; ++s) for (int k = 3; k >= 0; --k) y = std::fma(A[i * K + 4 * s + k], B[(4 * s + k) * 16 + j], y);
        for (int k = 0; k < K; ++k) { volatile double t = A[i * K + k] * B[k * 16 + j]; z = z + t; }
        for (int s = 0; s < nrep; ++s) { double t = 0; for (int k = 0; k < 4; ++k) t = std::fma(A[i * K + 4 * s + k], B[(4 * s + k) * 16 + j], t); p = p + t; }
        double g = D[i * 16 + j];
        asc += !memcmp(&g, &x, 8); desc += !memcmp(&g, &y, 8); mul += !memcmp(&g, &z, 8); pair += !memcmp(&g, &p, 8);
    }
    printf("bitwise matches of 256: ascending fma chain %d, descending %d, mul+add %d, block-then-add %d\n", asc, desc, mul, pair);
    // rates
    double *dout; hipMalloc(&dout, (size_t)256 * 8 * 1024 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpb : {4, 8, 16}) {
        const int iters = 20000, blocks = 256 * 2;
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(64 * wpb), 0, 0, dout, iters);
                if (mode == 1) hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(64 * wpb), 0, 0, dout, iters);
                if (mode == 2) hipLaunchKernelGGL(rate<2>, dim3(blocks), dim3(64 * wpb), 0, 0, dout, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per wave per iter: mfma 4 x 1024 fma; vector 64 fma-instr x 64 lanes = 4096 fma
            double fma = (double)blocks * wpb * iters * 4096.0;
            printf("waves/block %2d mode %d (%s): %.2f ms  %.1f TFLOP/s\n", wpb, mode, mode == 0 ? "mfma" : mode == 1 ? "v_fma" : "half/half", ms, 2 * fma / ms / 1e9);
        }
    }
    return 0;
}
