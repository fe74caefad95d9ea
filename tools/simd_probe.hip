// tools/simd_probe.hip -- which SIMD does wave w of a 1024-thread workgroup run on?  (HW_REG_HW_ID, gfx9: WAVE_ID [3:0], SIMD_ID [5:4], CU_ID [11:8])
//   hipcc --offload-arch=gfx950 -O2 tools/simd_probe.hip -o tools/_build/simd_probe && tools/_build/simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned *out)
{
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = hw;
}
int main()
{
    unsigned *d, h[4 * 16];
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 1;
    hipLaunchKernelGGL(probe, dim3(4), dim3(1024), 0, 0, d);
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int b = 0; b < 4; ++b) {
        printf("workgroup %d: wave -> simd:", b);
        for (int w = 0; w < 16; ++w) printf(" %d:%u", w, (h[b * 16 + w] >> 4) & 3);
        printf("   (cu %u)\n", (h[b * 16] >> 8) & 15);
    }
    return 0;
}
