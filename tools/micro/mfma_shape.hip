// Which fp32 MFMA shape sustains more FLOP/s under a conv-like load on MI355X?  (cdna_hip_programming.md rule 28 / DVFS
// give-back item 7: for bf16 the 16x16 shape held a higher clock than the 32x32 one.)  Each wave loops: 6 ds_read_b128 of
// random operands from LDS, then the same 32 x 128 x 8 worth of products either as 32 v_mfma_f32_32x32x2_f32 or as
// 64 v_mfma_f32_16x16x4_f32.  Two 256-thread workgroups per CU, random data, ~2 ms per launch.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shape.hip -o /tmp/mfma_shape && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k(const float *src, float *dst, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = src[(blockIdx.x * 8192 + i) & 0xFFFFF];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[8];
    f32x4 acs[32];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) acs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        const int base = ((it * 67 + lane) * 4) & 4095;
        f32x4 a[2], b[4];
#pragma unroll
        for (int m = 0; m < 2; ++m) a[m] = *reinterpret_cast<const f32x4 *>(lds + ((base + 512 * m) & 8188));
#pragma unroll
        for (int n = 0; n < 4; ++n) b[n] = *reinterpret_cast<const f32x4 *>(lds + ((base + 4096 + 384 * n) & 8188));
        if (SHAPE == 32) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[m * 4 + n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[n][j], a[m][j], acc[m * 4 + n], 0, 0, 0);
        } else {
            // the same FLOPs: 32 accumulators of 16x16, K = 4 per instruction -> two k-steps per operand quad pair
#pragma unroll
            for (int j = 0; j < 4; j += 2)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#pragma unroll
                        for (int h = 0; h < 4; ++h)
                            acs[(m * 4 + n) * 4 + h] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[n][j + (h & 1)], a[m][j + (h >> 1)], acs[(m * 4 + n) * 4 + h], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acs[i][0] + acs[i][1] + acs[i][2] + acs[i][3];
    dst[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    const int nb = 512, iters = 4000;
    std::vector<float> h(1 << 20);
    srand(1);
    for (auto &v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *src, *dst;
    hipMalloc(&src, h.size() * 4);
    hipMalloc(&dst, nb * 256 * 4);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int round = 0; round < 4; ++round)
        for (int shape : {32, 16}) {
            for (int w = 0; w < 3; ++w) {
                if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(nb), dim3(256), 0, 0, src, dst, iters);
                else hipLaunchKernelGGL(k<16>, dim3(nb), dim3(256), 0, 0, src, dst, iters);
            }
            hipEventRecord(e0);
            const int reps = 20;
            for (int r = 0; r < reps; ++r) {
                if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(nb), dim3(256), 0, 0, src, dst, iters);
                else hipLaunchKernelGGL(k<16>, dim3(nb), dim3(256), 0, 0, src, dst, iters);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            // per iteration per wave: 32 MFMAs x 32*32*2*2 FLOP
            const double flop = (double)reps * nb * 4 * iters * 32.0 * 4096.0;
            printf("round %d shape %dx%d: %.3f ms per launch, %.1f TFLOP/s\n", round, shape, shape, ms / reps, flop / (ms * 1e-3) / 1e12);
        }
    return 0;
}
