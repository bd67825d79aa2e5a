// What can the second wave of a SIMD get done beside a wave that streams v_mfma_f32_32x32x2_f32 and meets it at one s_barrier per
// 32 MFMAs?  One 512-thread workgroup per CU: waves 0-3 issue 32 MFMAs per step (2048 cycles of matrix pipe) and arrive at the
// barrier; waves 4-7 run a configurable instruction mix per step and arrive at the same barrier.  Reported: cycles per step with the
// MFMAs alone, with the mix alone (MFMA waves only at the barrier) and with both -- the last one should be max(first two) if the
// second wave's instructions are free for the matrix pipe and vice versa.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ws_step.hip -o tools/micro/ws_step && tools/micro/ws_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// (prio >> 4) == 3: the MFMA waves issue v_mfma_f32_32x32x16_bf16 instead (64 per step): is the serialisation a property of the fp32 instruction?
// prio bits: 1 = s_setprio 3 in the MFMA waves, 2 = in the second waves, 4 = stamps, 16/32 = MFMA accumulators (one: four dependent in a row, two), 256 = waves 4-7 issue
// the MFMAs and waves 0-3 the mix (the older wave of a SIMD is then the mixed one)
// mix bits: 1 = 6 ds_write_b128, 2 = 8 ds_read_b128 + use, 4 = 40 scalar instructions, 8 = 32 vector fmas, 16 = 2 buffer-like global loads
// (waited for in the NEXT step), 32 = 3 s_memtime, 64 = 16 ds_write_b128, 128 = 200 scalar instructions
template <int NACC, bool SWAP>
__global__ __launch_bounds__(512, 1) void k(const float *src, float *dst, long long *ticks, int steps, int mfma_on, int mix, int prio)
{
    __shared__ f32x4 lds[4096];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float a = src[lane], b = src[lane + 64];
    constexpr bool swap_roles = SWAP;
    constexpr int nacc = NACC;       // 0 = eight accumulators in rotation, 1 = one (every MFMA depends on the previous one), 2 = two
    if ((wave < 4) != swap_roles) {
        if (prio & 1) __builtin_amdgcn_s_setprio(3);
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        __syncthreads();
        const long long t0 = __builtin_readcyclecounter();
        long long work = 0, tl = t0;
        for (int s = 0; s < steps; ++s) {
            if (mfma_on) {
                if constexpr (nacc == 3) {      // the bf16 instruction (an XDL op, 32 cycles): 64 per step = the same 2048 cycles
                    bf16x8 ab;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ab[e] = (__bf16)a;
#pragma unroll
                    for (int q = 0; q < 64; ++q) acc[q & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc[q & 7], 0, 0, 0);
                } else if constexpr (nacc == 0) {
#pragma unroll
                    for (int q = 0; q < 32; ++q) acc[q & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q & 7], 0, 0, 0);
                } else if constexpr (nacc == 1) {
#pragma unroll
                    for (int q = 0; q < 32; ++q) acc[q >> 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q >> 2], 0, 0, 0);     // four in a row on one accumulator
                } else {
#pragma unroll
                    for (int q = 0; q < 32; ++q) acc[(q >> 3) * 2 + (q & 1)] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[(q >> 3) * 2 + (q & 1)], 0, 0, 0);   // two alternate
                }
            }
            if (prio & 4) { const long long ta = __builtin_readcyclecounter(); work += ta - tl; }
            if (!(prio & 8)) __syncthreads();
            if (prio & 4) tl = __builtin_readcyclecounter();
        }
        const long long t1 = __builtin_readcyclecounter();
        if (lane == 0) { ticks[blockIdx.x * 8 + wave] = t1 - t0; ticks[2048 + blockIdx.x * 8 + wave] = work; }
        float sum = 0.f;
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
        dst[blockIdx.x * 512 + threadIdx.x] = sum;
        return;
    }
    if (prio & 2) __builtin_amdgcn_s_setprio(3);
    f32x4 v = {a, b, a + 1.f, b + 1.f}, ld0 = v, ld1 = v;
    int sc = blockIdx.x;
    float x0 = a, x1 = b, x2 = a + 1.f, x3 = b + 1.f;
    long long tm = 0;
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(src);
    __syncthreads();
    long long work = 0, tl = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
        if (mix & 16) {
            v += ld0 + ld1;               // waits for the loads of the previous step
            ld0 = g4[(lane + s * 64) & 1023];
            ld1 = g4[(lane + s * 64 + 512) & 1023];
        }
        if (mix & 1) {
#pragma unroll
            for (int i = 0; i < 6; ++i) lds[(wave & 3) * 1024 + i * 64 + lane] = v;
        }
        if (mix & 64) {
#pragma unroll
            for (int i = 0; i < 16; ++i) lds[(wave & 3) * 1024 + i * 64 + lane] = v;
        }
        if (mix & 2) {
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 8; ++i) r += lds[(wave & 3) * 1024 + (8 + i) * 64 + lane];
            v += r;
        }
        if (mix & 4) {
#pragma unroll
            for (int i = 0; i < 40; ++i) asm volatile("s_add_i32 %0, %0, 3" : "+s"(sc));
        }
        if (mix & 128) {
#pragma unroll
            for (int i = 0; i < 200; ++i) asm volatile("s_add_i32 %0, %0, 3" : "+s"(sc));
        }
        if (mix & 8) {
            for (int rep = 0; rep < ((mix & 256) ? 10 : 1); ++rep)
#pragma unroll
            for (int i = 0; i < 8; ++i) { x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f); }
        }
        if (mix & 32) {
            tm += __builtin_readcyclecounter();
            tm ^= __builtin_readcyclecounter();
            tm += __builtin_readcyclecounter();
        }
        if (prio & 4) { const long long ta = __builtin_readcyclecounter(); work += ta - tl; }
        if (!(prio & 8)) __syncthreads();
        if (prio & 4) tl = __builtin_readcyclecounter();
    }
    if (prio & 8) work = __builtin_readcyclecounter() - tl;
    if (lane == 0) ticks[2048 + blockIdx.x * 8 + wave] = work;
    dst[blockIdx.x * 512 + threadIdx.x] = v[0] + v[1] + v[2] + v[3] + x0 + x1 + x2 + x3 + (float)sc + (float)tm + ld0[0] + ld1[0];
}

int main()
{
    float *src, *dst;
    long long *ticks;
    (void)hipMalloc(&src, 1 << 20);
    (void)hipMalloc(&dst, 256 * 512 * 4);
    (void)hipMalloc(&ticks, 2 * 256 * 8 * 8);
    (void)hipMemset(src, 0, 1 << 20);
    const int steps = 400, nb = 256;
    struct { int mix; const char *what; } mixes[] = {{0, "nothing"}, {64, "16 ds_write_b128"}, {2, "8 ds_read_b128"}, {128, "200 scalar adds"},
                                                     {8, "32 vector fmas"}, {8 | 256, "320 vector fmas"}, {16, "2 global loads (used a step later)"}, {1 | 2 | 4 | 8 | 16, "writes+reads+scalar+fmas+loads"}};
    for (int prio : {0, 48, 8, 8 + 48, 8 + 16, 8 + 256})
        for (auto &m : mixes) {
            double per[2] = {0, 0}, wc[2] = {0, 0}, wp[2] = {0, 0};
            for (int mf = 0; mf < 2; ++mf) {
                (void)hipMemset(ticks, 0, 2 * nb * 8 * 8);
                switch (prio >> 4) {
                    case 0: hipLaunchKernelGGL((k<0, false>), dim3(nb), dim3(512), 0, 0, src, dst, ticks, steps, mf, m.mix, prio); break;
                    case 1: hipLaunchKernelGGL((k<1, false>), dim3(nb), dim3(512), 0, 0, src, dst, ticks, steps, mf, m.mix, prio); break;
                    case 2: hipLaunchKernelGGL((k<2, false>), dim3(nb), dim3(512), 0, 0, src, dst, ticks, steps, mf, m.mix, prio); break;
                    case 3: hipLaunchKernelGGL((k<3, false>), dim3(nb), dim3(512), 0, 0, src, dst, ticks, steps, mf, m.mix, prio); break;
                    case 16: hipLaunchKernelGGL((k<0, true>), dim3(nb), dim3(512), 0, 0, src, dst, ticks, steps, mf, m.mix, prio); break;
                    case 17: hipLaunchKernelGGL((k<1, true>), dim3(nb), dim3(512), 0, 0, src, dst, ticks, steps, mf, m.mix, prio); break;
                    default: hipLaunchKernelGGL((k<2, true>), dim3(nb), dim3(512), 0, 0, src, dst, ticks, steps, mf, m.mix, prio); break;
                }
                (void)hipDeviceSynchronize();
                std::vector<long long> h(2 * nb * 8);
                (void)hipMemcpy(h.data(), ticks, 2 * nb * 8 * 8, hipMemcpyDeviceToHost);
                double s = 0, c = 0, q = 0; int n = 0;
                const int o = (prio & 256) ? 4 : 0;
                for (int b = 0; b < nb; ++b) for (int w = 0; w < 4; ++w) { s += (double)h[b * 8 + o + w]; c += (double)h[2048 + b * 8 + o + w]; q += (double)h[2048 + b * 8 + 4 - o + w]; ++n; }
                per[mf] = s / n / steps; wc[mf] = c / n / steps; wp[mf] = q / n / steps;
            }
            printf("prio %3d  second wave per step: %-36s  cycles per step: alone %7.1f (its work %6.1f)   beside 32 MFMAs %7.1f (MFMA wave busy %6.1f, second wave's work %6.1f)\n",
                   prio, m.what, per[0], wp[0], per[1], wc[1], wp[1]);
        }
    return 0;
}
