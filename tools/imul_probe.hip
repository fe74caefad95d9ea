// tools/imul_probe.hip -- issue cost of the instructions the pinned normal generator is made of, one wave per SIMD and two (gfx950):
// v_mad_u64_u32 (Philox4x32's 32 x 32 -> 64 products: twenty per block), v_mul_hi_u32 + v_mul_lo_u32 (the two-instruction form),
// v_fma_f64, v_rcp_f64 / v_rsq_f64 (inside the correctly rounded division and square root), and one whole Philox block.
//   hipcc --offload-arch=gfx950 -O2 tools/imul_probe.hip -o tools/_probe_imul && tools/_probe_imul
// Each kernel runs N dependent-free groups of eight independent chains (so the pipe, not the latency, is what is measured) and reports
// cycles per wave-instruction from s_memtime (constant 100 MHz on this part: scaled by the measured core clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int REP = 4096;

template <int KIND>
__global__ __launch_bounds__(64) void probe(uint64_t *out, uint32_t seed)
{
    uint32_t a[8]; uint64_t acc[8]; double f[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + 2654435761u * (threadIdx.x + 64 * i + 1); acc[i] = a[i]; f[i] = 1.0 + 1e-9 * (double)a[i]; }
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) {                                  // v_mad_u64_u32 (the compiler's form of a 32 x 32 -> 64 product)
                uint64_t p;
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p) : "v"(a[i]), "v"(0xD2511F53u) : "vcc");
                a[i] = (uint32_t)(p >> 32) ^ (uint32_t)p;
            } else if (KIND == 1) {                           // v_mul_hi_u32 + v_mul_lo_u32
                uint32_t hi, lo;
                asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(hi) : "v"(a[i]), "v"(0xD2511F53u));
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(lo) : "v"(a[i]), "v"(0xD2511F53u));
                a[i] = hi ^ lo;
            } else if (KIND == 2) {                           // v_fma_f64
                asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f[i]) : "v"(0.999999));
            } else if (KIND == 3) {                           // v_rcp_f64
                asm volatile("v_rcp_f64 %0, %0" : "+v"(f[i]));
            } else if (KIND == 4) {                           // v_rsq_f64
                asm volatile("v_rsq_f64 %0, %0" : "+v"(f[i]));
            } else if (KIND == 5) {                           // v_xor_b32 (the full-rate reference)
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            } else if (KIND == 6) {                           // v_mul_u32_u24
                asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(0x511F53u));
            } else if (KIND == 7) {                           // v_cvt_f64_u32
                asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(f[i]) : "v"(a[i]));
            }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    uint64_t s = 0; double fs = 0;
    for (int i = 0; i < 8; ++i) { s += a[i] + acc[i]; fs += f[i]; }
    if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;
    if (s == 0x1234567u && fs == 1.5) out[blockIdx.x * 2 + 1] = s;         // keep the results alive
}

template <int KIND>
static int run(const char *name, int per_group, int blocks)
{
    uint64_t *d;
    CHK(hipMalloc(&d, sizeof(uint64_t) * 2 * blocks));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(64), 0, 0, d, 12345u);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(64), 0, 0, d, 12345u);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> h(2 * blocks);
    CHK(hipMemcpy(h.data(), d, sizeof(uint64_t) * 2 * blocks, hipMemcpyDeviceToHost));
    double ticks = 0; for (int b = 0; b < blocks; ++b) ticks += (double)h[2 * b]; ticks /= blocks;
    const double n = (double)REP * 8 * per_group;
    // wall time of the launch / instructions of one wave, in core cycles at 2.4 GHz (waves per SIMD = blocks / 1024)
    printf("%-34s %5d waves: %7.2f counter ticks per wave-instruction; launch %.3f ms = %6.2f cycles(2.4 GHz) per wave-instruction per SIMD-resident wave\n",
           name, blocks, ticks / n, ms, ms * 1e-3 * 2.4e9 / n / ((blocks + 1023) / 1024));
    CHK(hipFree(d));
    return 0;
}

int main()
{
    for (int blocks : {1024, 2048}) {
        if (run<5>("v_xor_b32", 1, blocks)) return 1;
        if (run<0>("v_mad_u64_u32 (+ 1 xor)", 1, blocks)) return 1;
        if (run<1>("v_mul_hi_u32 + v_mul_lo_u32 (+ xor)", 1, blocks)) return 1;
        if (run<6>("v_mul_u32_u24", 1, blocks)) return 1;
        if (run<2>("v_fma_f64", 1, blocks)) return 1;
        if (run<7>("v_cvt_f64_u32", 1, blocks)) return 1;
        if (run<3>("v_rcp_f64", 1, blocks)) return 1;
        if (run<4>("v_rsq_f64", 1, blocks)) return 1;
    }
    return 0;
}
