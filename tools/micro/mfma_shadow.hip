// Does vector work of the SAME wave run in the shadow of its matrix instructions?  One wave per SIMD (one workgroup of 256 threads per
// CU), a loop of 16 v_mfma_f32_32x32x16_bf16 (four accumulators in rotation, as conv_wino_b3's planes) and 0 / 96 independent vector
// instructions, (a) grouped behind the MFMAs, (b) interleaved 6 per MFMA, (c) interleaved and WRITING the registers the next MFMAs
// read as their B operand (what a transform + split does).  Prints wave cycles per loop iteration; 16 MFMAs alone are 16 x 32 = 512.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shadow.hip -o tools/micro/mfma_shadow && tools/micro/mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NV>     // MODE 0 grouped, 1 interleaved, 2 interleaved + the vector results feed the next iteration's B operands
__global__ __launch_bounds__(256) void k(float *out, const float *in, int iters, unsigned long long *cyc)
{
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    u32x4 A = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    u32x4 B[4];
    for (int j = 0; j < 4; ++j) B[j] = u32x4{0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u + (unsigned)j};
    float v[24];
    for (int i = 0; i < 24; ++i) v[i] = in[(threadIdx.x + i) & 255];
    const float c0 = in[3], c1 = in[5];
    const unsigned long long t0 = __builtin_readcyclecounter();
    // every instruction of the loop body is a volatile asm statement: the order below is the order in the binary
#define MFMA(m) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[(m) & 3]) : "v"(A), "v"(B[(m) & 3]))
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i) % 24]) : "v"(c0), "v"(c1))
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int m = 0; m < 16; ++m) MFMA(m);
#pragma unroll
            for (int q = 0; q < NV; ++q) FMA(q);
        } else {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                MFMA(m);
#pragma unroll
                for (int q = 0; q < NV / 16; ++q) {
                    FMA(m * (NV / 16) + q);
                    if (MODE == 2 && q == 0)        // one vector result per group becomes part of the B operand of the MFMA two slots on
                        asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(B[(m + 2) & 3][m & 3]) : "v"(v[(m * (NV / 16)) % 24]), "v"(0x3fff3fffu), "v"(0x3f003f00u));
                }
            }
        }
    }
#undef MFMA
#undef FMA
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 24; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NV>
static void run(const char *what, float *out, float *in, unsigned long long *cyc, int cus)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(cus), dim3(256), 0, 0, out, in, 100, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(cus), dim3(256), 0, 0, out, in, iters, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(cus);
    hipMemcpy(h.data(), cyc, cus * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto c : h) mean += (double)c;
    mean /= cus;
    // s_memtime / readcyclecounter ticks at 100 MHz: convert through the wall time
    const double us_per_iter = ms * 1e3 / iters;
    printf("%-78s %7.3f us / iteration = %6.0f cycles at 2.0 GHz  (16 MFMAs alone: 512)   %.3f ms\n", what, us_per_iter, us_per_iter * 2000.0, ms);
}

int main()
{
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    float *out, *in;
    unsigned long long *cyc;
    hipMalloc(&out, cus * 256 * 4); hipMalloc(&in, 1024); hipMalloc(&cyc, cus * 8);
    std::vector<float> h(256, 0.5f);
    hipMemcpy(in, h.data(), 1024, hipMemcpyHostToDevice);
    run<0, 0>("16 MFMAs alone", out, in, cyc, cus);
    run<0, 96>("16 MFMAs, then 96 fmas (grouped)", out, in, cyc, cus);
    run<1, 96>("16 x (MFMA, 6 fmas) interleaved", out, in, cyc, cus);
    run<1, 64>("16 x (MFMA, 4 fmas) interleaved", out, in, cyc, cus);
    run<1, 32>("16 x (MFMA, 2 fmas) interleaved", out, in, cyc, cus);
    run<2, 96>("16 x (MFMA, 6 fmas) interleaved, one result per group feeds a later MFMA's B operand", out, in, cyc, cus);
    run<0, 192>("16 MFMAs, then 192 fmas (grouped)", out, in, cyc, cus);
    run<1, 192>("16 x (MFMA, 12 fmas) interleaved", out, in, cyc, cus);
    return 0;
}
