// Back-to-back launch cost of an empty kernel as a function of workgroup size, dynamic LDS and grid (what a launch costs before
// it does anything): hipcc --offload-arch=gfx950 -O3 tools/micro/empty_launch.hip -o /tmp/empty_launch && /tmp/empty_launch
#include <hip/hip_runtime.h>
#include <cstdio>
extern "C" __global__ void empty_k(int *p) { extern __shared__ int sm[]; if (p && threadIdx.x == 9999) p[0] = sm[0]; }
int main()
{
    hipFuncSetAttribute(reinterpret_cast<const void *>(empty_k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int cfg[][3] = {{256, 1024, 160 * 1024}, {256, 1024, 80 * 1024}, {256, 1024, 0}, {256, 512, 160 * 1024}, {256, 512, 0}, {256, 256, 0},
                          {512, 512, 72 * 1024}, {2048, 512, 72 * 1024}, {2048, 256, 0}, {256, 768, 160 * 1024}};
    for (auto &c : cfg) {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(empty_k, dim3(c[0]), dim3(c[1]), c[2], 0, nullptr);
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_k, dim3(c[0]), dim3(c[1]), c[2], 0, nullptr);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("grid %5d x %4d threads, %6d B LDS: %.2f us per launch\n", c[0], c[1], c[2], ms * 1000 / 200);
    }
    return 0;
}
