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
    for (int it = 0; MODE < 3 && it < iters; ++it) {
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
#undef FMA
    // MODE 3: the split's own instruction mix, 7 per MFMA and independent of each other: 2 v_cvt_pk_bf16_f32, 2 shifts, 2 ands, 1 v_pk_add_f32
    // MODE 4: seven v_cvt_pk_bf16_f32; MODE 5: seven v_pk_add_f32; MODE 6: seven v_lshlrev_b32; 7: v_dot2_f32_bf16; 8: v_sub_f32; 9: v_perm_b32
    if (MODE >= 3) {
        float2 pk[4] = {{v[0], v[1]}, {v[2], v[3]}, {v[4], v[5]}, {v[6], v[7]}};
        unsigned w[8];
        for (int i = 0; i < 8; ++i) w[i] = __builtin_bit_cast(unsigned, v[8 + i]);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                MFMA(m);
                // every instruction updates a register of its own in place (results stay live: dead outputs would all be given one
                // register and the compiler would put an s_nop between two asm statements that write it)
                if (MODE == 3) {
                    asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(w[0]) : "v"(v[17]));
                    asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(w[1]));
                    asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(w[2]));
                    asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(pk[m & 3]) : "v"(pk[(m + 2) & 3]));
                    asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(w[3]) : "v"(v[19]));
                    asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(w[6]));
                    asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(w[7]));
                } else {
#pragma unroll
                    for (int q = 0; q < 7; ++q) {
                        if (MODE == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(w[q]) : "v"(v[17]));
                        if (MODE == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[q & 3]) : "v"(pk[(q + 2) & 3]));
                        if (MODE == 6) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(w[q]));
                        if (MODE == 7) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(v[q]) : "v"(w[7]), "v"(0x0000bf80u));
                        if (MODE == 8) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[q]) : "v"(v[17]));
                        if (MODE == 9) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(w[q]) : "v"(w[7]), "v"(0x07060302u));
                    }
                }
            }
        }
        for (int i = 0; i < 8; ++i) v[i] += __builtin_bit_cast(float, w[i]);
        for (int i = 0; i < 4; ++i) v[8 + i] += pk[i].x + pk[i].y;
    }
#undef MFMA
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
    // the kernel's own counter (s_memtime: shader clocks on this part -- 2.07e9 per second of a conv_wino_b3 launch) beside the wall time
    const double us_per_iter = ms * 1e3 / iters;
    printf("%-78s %7.3f us / iteration, %6.0f counter ticks / iteration (%.2f GHz if they are clocks; 16 MFMAs alone: 512)\n", what, us_per_iter,
           mean / iters, mean / iters / us_per_iter / 1e3);
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
    run<3, 0>("16 x (MFMA, the split's mix: 2 cvt_pk_bf16, 2 shifts, 2 ands, 1 pk_add_f32)", out, in, cyc, cus);
    run<4, 0>("16 x (MFMA, 7 v_cvt_pk_bf16_f32)", out, in, cyc, cus);
    run<5, 0>("16 x (MFMA, 7 v_pk_add_f32)", out, in, cyc, cus);
    run<6, 0>("16 x (MFMA, 7 v_lshlrev_b32)", out, in, cyc, cus);
    run<7, 0>("16 x (MFMA, 7 v_dot2_f32_bf16)", out, in, cyc, cus);
    run<8, 0>("16 x (MFMA, 7 v_sub_f32)", out, in, cyc, cus);
    run<9, 0>("16 x (MFMA, 7 v_perm_b32)", out, in, cyc, cus);
    return 0;
}
