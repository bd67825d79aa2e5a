// How many independent accumulators does v_mfma_f32_16x16x4_f32 need to run at its issue rate?  Each wave issues the
// instruction round-robin over NACC accumulators from register operands (no memory in the loop); cycles per instruction from
// s_memtime, one or two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_chain.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC, int SHAPE>
__global__ __launch_bounds__(256, 2) void k(const float *src, float *dst, long long *ticks, int iters)
{
    const int lane = threadIdx.x & 63;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = src[(lane + 64 * i) & 1023]; b[i] = src[(lane * 3 + 64 * i + 7) & 1023]; }
    f32x4 acc[NACC];
    f32x16 big[NACC > 8 ? 1 : NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < (NACC > 8 ? 1 : NACC); ++i)
        for (int r = 0; r < 16; ++r) big[i][r] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (SHAPE == 16) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[(q + i) & 7], acc[i], 0, 0, 0);
                else big[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[(q + i) & 7], big[i], 0, 0, 0);
            }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < (NACC > 8 ? 1 : NACC); ++i)
        for (int r = 0; r < 16; ++r) s += big[i][r];
    dst[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int NACC, int SHAPE>
static void run(const float *src, float *dst, long long *ticks, int nb, const char *what)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, SHAPE>), dim3(nb), dim3(256), 0, 0, src, dst, ticks, iters);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NACC, SHAPE>), dim3(nb), dim3(256), 0, 0, src, dst, ticks, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 8 * NACC;
    const double flop = n * (SHAPE == 16 ? 2048.0 : 4096.0) * 4 * nb;
    printf("%-28s NACC %2d: %7.1f us  %6.1f TFLOP/s  %5.1f ns per instruction and wave\n", what, NACC, ms * 1e3, flop / ms / 1e9, ms * 1e6 / n);
}

int main()
{
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 37) % 101) / 101.f - 0.5f;
    float *src, *dst;
    long long *ticks;
    (void)hipMalloc(&src, 4096); (void)hipMalloc(&dst, 1024 * 256 * 4); (void)hipMalloc(&ticks, 1024 * 8);
    (void)hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
    run<1, 16>(src, dst, ticks, 512, "16x16x4, 2 waves per SIMD");
    run<2, 16>(src, dst, ticks, 512, "16x16x4, 2 waves per SIMD");
    run<4, 16>(src, dst, ticks, 512, "16x16x4, 2 waves per SIMD");
    run<7, 16>(src, dst, ticks, 512, "16x16x4, 2 waves per SIMD");
    run<16, 16>(src, dst, ticks, 512, "16x16x4, 2 waves per SIMD");
    run<7, 16>(src, dst, ticks, 256, "16x16x4, 1 wave per SIMD");
    run<16, 16>(src, dst, ticks, 256, "16x16x4, 1 wave per SIMD");
    run<1, 32>(src, dst, ticks, 512, "32x32x2, 2 waves per SIMD");
    run<2, 32>(src, dst, ticks, 512, "32x32x2, 2 waves per SIMD");
    run<4, 32>(src, dst, ticks, 512, "32x32x2, 2 waves per SIMD");
    run<4, 32>(src, dst, ticks, 256, "32x32x2, 1 wave per SIMD");
    run<2, 32>(src, dst, ticks, 256, "32x32x2, 1 wave per SIMD");
    run<1, 32>(src, dst, ticks, 256, "32x32x2, 1 wave per SIMD");
    run<1, 16>(src, dst, ticks, 256, "16x16x4, 1 wave per SIMD");
    run<2, 16>(src, dst, ticks, 256, "16x16x4, 1 wave per SIMD");
    run<4, 16>(src, dst, ticks, 256, "16x16x4, 1 wave per SIMD");
    return 0;
}
