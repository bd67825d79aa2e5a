// Where do the waves of a 256-thread workgroup land?  Every wave records HW_ID (SIMD, CU, SE) and XCC_ID.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/wave_simd.hip -o /tmp/wave_simd && /tmp/wave_simd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 2) void k(unsigned *out, int spin)
{
    __shared__ float pad[4];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float a = threadIdx.x;
    for (int i = 0; i < spin * ((threadIdx.x >> 6) == 3 ? 1 : 8); ++i) a = a * 1.0001f + 0.5f;     // wave 3 is the light one
    if (a == 123.f) pad[0] = a;
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
}
int main()
{
    const int nb = 4096;
    unsigned *d;
    hipMalloc(&d, nb * 4 * 2 * 4);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, 2000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 8);
    hipMemcpy(h.data(), d, nb * 32, hipMemcpyDeviceToHost);
    int hist[4][4] = {};
    for (int b = 0; b < nb; ++b)
        for (int w = 0; w < 4; ++w) hist[w][(h[(b * 4 + w) * 2] >> 4) & 3]++;
    for (int w = 0; w < 4; ++w) printf("wave %d -> SIMD 0..3: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    for (int b : {0, 1, 8, 9, 256, 257, 2048, 4000}) {
        printf("block %4d:", b);
        for (int w = 0; w < 4; ++w) {
            const unsigned v = h[(b * 4 + w) * 2];
            printf("  w%d simd %u cu %u se %u xcc %u |", w, (v >> 4) & 3, (v >> 8) & 15, (v >> 13) & 7, h[(b * 4 + w) * 2 + 1] & 15);
        }
        printf("\n");
    }
    return 0;
}
