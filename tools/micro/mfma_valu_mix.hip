// How much vector work fits beside v_mfma_f32_32x32x2_f32 before the matrix pipe starves?  (conv_wino.hip computes the B operand of
// every MFMA with ~2-3 fp32 adds.)  Each wave loops: 32 MFMAs into 8 accumulators; NV VALU instructions per MFMA; DEP = 1: the
// MFMA's B operand is the result of those instructions (a chain of fp32 adds), DEP = 0: the VALU work is an independent chain.
// Two 256-thread workgroups per CU (2 waves per SIMD) or one (WPS = 1).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_mix.hip -o /tmp/mfma_valu_mix && /tmp/mfma_valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NV, int DEP>
__global__ __launch_bounds__(256, 2) void k(const float *src, float *dst, int iters)
{
    const int lane = threadIdx.x & 63;
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[8], x[8], y = src[lane + 64];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = src[(blockIdx.x * 64 + lane + i * 64) & 0xFFFF]; x[i] = a[i] * 0.5f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float bop = x[i];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    if (DEP) bop = bop + a[(i + v + 1) & 7];
                    else { y = y + a[(i + v) & 7]; asm volatile("" : "+v"(y)); }
                }
                if (DEP) asm volatile("" : "+v"(bop));
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bop, acc[i], 0, 0, 0);
            }
    }
    float s = y;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    dst[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int DEP>
static void run(const float *src, float *dst, int nb, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<NV, DEP>), dim3(nb), dim3(256), 0, 0, src, dst, iters);
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<NV, DEP>), dim3(nb), dim3(256), 0, 0, src, dst, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)reps * nb * 4 * iters * 32.0 * 4096.0;
    printf("workgroups %d (%d per CU)  VALU per MFMA %d  %s: %.3f ms per launch, %.1f TFLOP/s\n", nb, nb / 256, NV, DEP ? "feeding the MFMA operand" : "independent chain        ",
           ms / reps, flop / (ms * 1e-3) / 1e12);
}

int main()
{
    const int iters = 2000;
    std::vector<float> h(1 << 16);
    srand(1);
    for (auto &v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *src, *dst;
    hipMalloc(&src, h.size() * 4);
    hipMalloc(&dst, 512 * 256 * 4);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int nb : {512, 256}) {
        run<0, 0>(src, dst, nb, iters);
        run<1, 1>(src, dst, nb, iters);
        run<2, 1>(src, dst, nb, iters);
        run<3, 1>(src, dst, nb, iters);
        run<4, 1>(src, dst, nb, iters);
        run<6, 1>(src, dst, nb, iters);
        run<8, 1>(src, dst, nb, iters);
        run<2, 0>(src, dst, nb, iters);
        run<4, 0>(src, dst, nb, iters);
        run<8, 0>(src, dst, nb, iters);
        run<12, 0>(src, dst, nb, iters);
    }
    return 0;
}
