// What does a wave that issues fp32 MFMAs back to back do to its neighbour on the SIMD?  One 512-thread workgroup per CU = two waves
// per SIMD: waves 0-3 run MFMAs only (32x32x2: 64-cycle slots, or 16x16x4: 32-cycle slots, the same FLOPs), waves 4-7 a
// chain of vector instructions (a dependent fma chain, or independent fmas) and report how long it took -- grouped by what the other
// wave of their SIMD (HW_ID) was doing: nothing, MFMAs, or another vector chain.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_neighbour.hip -o tools/micro/mfma_neighbour && tools/micro/mfma_neighbour
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int SHAPE, bool DEP>
__global__ __launch_bounds__(512, 1) void k(const float *src, float *dst, long long *ticks, int iters, int mfma_on, int valu_iters)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const bool valu_wg = wave >= 4;          // waves w and w + 4 of a workgroup land on the same SIMD
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (lane == 0) ticks[(blockIdx.x * 8 + wave) * 2 + 1] = (long long)((hw >> 4) & 3) | (valu_wg ? 16 : 0);
    float a = src[lane], b = src[lane + 64];
    if (!valu_wg) {
        if (!mfma_on) return;
        f32x16 big[2];
        f32x4 sm[4];
        for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) big[i][r] = 0.f;
        for (int i = 0; i < 4; ++i) sm[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const long long m0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (SHAPE == 32) big[q & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[q & 1], 0, 0, 0);
                else { sm[q & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, sm[q & 3], 0, 0, 0); sm[(q + 2) & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, sm[(q + 2) & 3], 0, 0, 0); }
            }
        }
        const long long m1 = __builtin_readcyclecounter();
        if (lane == 0) ticks[(blockIdx.x * 8 + wave) * 2] = m1 - m0;
        float s = 0.f;
        for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += big[i][r];
        for (int i = 0; i < 4; ++i) s += sm[i][0] + sm[i][1] + sm[i][2] + sm[i][3];
        dst[blockIdx.x * 512 + threadIdx.x] = s;
        return;
    }
    // the vector workgroup: wait a little so that the MFMA neighbours are running, then time 4096 vector instructions
    for (int i = 0; i < 2000; ++i) a = a * 1.0001f + 0.25f;
    const long long t0 = __builtin_readcyclecounter();
    float x0 = a, x1 = b, x2 = a + 1.f, x3 = b + 1.f, x4 = a + 2.f, x5 = b + 2.f, x6 = a + 3.f, x7 = b + 3.f;
    for (int i = 0; i < valu_iters; ++i) {
        if (DEP) { x0 = fmaf(x0, 1.0001f, 0.5f); x0 = fmaf(x0, 1.0001f, 0.5f); x0 = fmaf(x0, 1.0001f, 0.5f); x0 = fmaf(x0, 1.0001f, 0.5f);
                   x0 = fmaf(x0, 1.0001f, 0.5f); x0 = fmaf(x0, 1.0001f, 0.5f); x0 = fmaf(x0, 1.0001f, 0.5f); x0 = fmaf(x0, 1.0001f, 0.5f); }
        else { x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f);
               x4 = fmaf(x4, 1.0001f, 0.5f); x5 = fmaf(x5, 1.0001f, 0.5f); x6 = fmaf(x6, 1.0001f, 0.5f); x7 = fmaf(x7, 1.0001f, 0.5f); }
    }
    const long long t1 = __builtin_readcyclecounter();
    dst[blockIdx.x * 512 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (lane == 0) ticks[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
}

template <int SHAPE, bool DEP>
static void run(const float *src, float *dst, long long *ticks, int mfma_on, const char *what, int valu_iters = 512)
{
    const int nb = 256;
    (void)hipMemset(ticks, 0, nb * 8 * 2 * 8);
    hipLaunchKernelGGL((k<SHAPE, DEP>), dim3(nb), dim3(512), 0, 0, src, dst, ticks, 400, mfma_on, valu_iters);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(nb * 16);
    (void)hipMemcpy(h.data(), ticks, nb * 16 * 8, hipMemcpyDeviceToHost);
    double s[2] = {0, 0}; int n[2] = {0, 0};
    for (int b = 0; b < nb; ++b)
        for (int w = 4; w < 8; ++w) {
            const int simd = (int)(h[(b * 8 + w) * 2 + 1] & 3);
            bool beside_mfma = false;
            for (int v = 0; v < 4; ++v) beside_mfma |= (int)(h[(b * 8 + v) * 2 + 1] & 3) == simd;
            s[beside_mfma] += (double)h[(b * 8 + w) * 2] * 512.0 / valu_iters; n[beside_mfma]++;
        }
    double ms = 0; int mn = 0;
    for (int b = 0; b < nb; ++b) for (int w = 0; w < 4; ++w) if (h[(b * 8 + w) * 2] > 0) { ms += (double)h[(b * 8 + w) * 2]; ++mn; }
    if (mn) printf("    MFMA waves: %.1f cycles per %s (n=%d)\n", ms / mn / (400.0 * 16), SHAPE == 32 ? "32x32x2 instruction" : "pair of 16x16x4 instructions", mn);
    printf("%-44s SIMD partner is a vector wave: %6.1f cycles per instruction (n=%d);  an MFMA wave%s: %6.1f (n=%d)\n", what,
           n[0] ? s[0] / n[0] / 4096.0 : 0.0, n[0], mfma_on ? "" : " (idle here)", n[1] ? s[1] / n[1] / 4096.0 : 0.0, n[1]);
}

int main()
{
    std::vector<float> h(256, 0.5f);
    float *src, *dst;
    long long *ticks;
    (void)hipMalloc(&src, 1024); (void)hipMalloc(&dst, 256 * 512 * 4); (void)hipMalloc(&ticks, 256 * 16 * 8);
    (void)hipMemcpy(src, h.data(), 1024, hipMemcpyHostToDevice);
    run<32, true>(src, dst, ticks, 0, "dependent fma chain, MFMA waves idle");
    run<32, true>(src, dst, ticks, 1, "dependent fma chain, 32x32x2 MFMAs");
    run<16, true>(src, dst, ticks, 1, "dependent fma chain, 16x16x4 MFMAs");
    run<32, false>(src, dst, ticks, 0, "independent fmas, MFMA waves idle");
    run<32, false>(src, dst, ticks, 1, "independent fmas, 32x32x2 MFMAs");
    run<16, false>(src, dst, ticks, 1, "independent fmas, 16x16x4 MFMAs");
    printf("-- the vector chains as long as the MFMA loops (the MFMA waves' own pace, first with a short chain beside them):\n");
    run<32, false>(src, dst, ticks, 1, "independent fmas x 40, 32x32x2 MFMAs", 512 * 40);
    run<32, true>(src, dst, ticks, 1, "dependent chain x 20, 32x32x2 MFMAs", 512 * 20);
    run<16, false>(src, dst, ticks, 1, "independent fmas x 40, 16x16x4 MFMAs", 512 * 40);
    return 0;
}
