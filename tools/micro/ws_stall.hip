// Which instructions of the second wave of a SIMD stall until its partner's v_mfma_f32_32x32x2_f32 stream pauses?
// One 512-thread workgroup per CU, no barriers: waves 0-3 issue blocks of NM back-to-back MFMAs (one scalar loop branch between
// blocks), waves 4-7 run `iters` iterations of a probe: 16 independent v_fma_f32 plus ONE instruction of the kind under test.
// Reported: cycles per probe iteration alone (MFMA waves idle) and beside the MFMA stream, for MFMA blocks of 8, 32 and 128, gapless
// or with one / three `s_nop 15` (64 cycles each) behind every MFMA; for the plain probe also the VGPR base of each wave (the
// starvation is not a matter of where the registers were allocated).
// What it shows (round 5): beside a gapless stream the probe usually keeps its pace (the mfma_neighbour result) -- but not in every
// launch; with ONE s_nop 15 behind each MFMA (the wave asks for nothing for 64 of every 68 cycles) the probe does not run at all
// until the stream ends; with three (matrix pipe idle two thirds of the time) it runs at 1.5 x its time alone.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ws_stall.hip -o tools/micro/ws_stall && tools/micro/ws_stall
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NM, int KIND, int GAP>
__global__ __launch_bounds__(512, 1) void k(const float *src, float *dst, long long *ticks, int blocks, int mfma_on, int iters)
{
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float a = src[lane], b = src[lane + 64];
    {   // where the wave's registers are: HW_REG_GPR_ALLOC (base and size in allocation granules) and HW_ID (SIMD)
        unsigned ga, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_GPR_ALLOC)" : "=s"(ga));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        if (lane == 0) ticks[2048 + blockIdx.x * 8 + wave] = ((long long)ga << 32) | hw;
    }
    if (wave < 4) {
        if (!mfma_on) return;
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const long long t0 = __builtin_readcyclecounter();
        for (int s = 0; s < blocks; ++s) {
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                acc[q & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q & 7], 0, 0, 0);
                if (GAP > 0) {       // the MFMA wave asks for nothing for 16 GAP cycles: s_nop 15, GAP times
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int gq = 0; gq < GAP; ++gq) asm volatile("s_nop 15");
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        const long long t1 = __builtin_readcyclecounter();
        if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
        float sum = 0.f;
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
        dst[blockIdx.x * 512 + threadIdx.x] = sum;
        return;
    }
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = a + (float)i;
    int sacc = 0;
    unsigned vint = (unsigned)lane;
    for (int i = 0; i < 3000; ++i) a = a * 1.0001f + 0.25f;      // let the MFMA stream start
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = fmaf(x[i], 1.0001f, 0.5f);
        if (KIND == 1) {                       // v_cmp -> SGPR pair, read by a scalar instruction
            unsigned long long m;
            asm volatile("v_cmp_gt_f32_e64 %0, %1, %2\n\ts_nop 4" : "=s"(m) : "v"(x[0]), "v"(b));
            asm volatile("s_add_u32 %0, %0, %1" : "+s"(sacc) : "s"((unsigned)m));
        } else if (KIND == 2) {                // v_cmp -> VCC, read by v_cndmask
            asm volatile("v_cmp_gt_f32_e32 vcc, %1, %2\n\tv_cndmask_b32_e32 %0, %1, %2, vcc" : "+v"(x[1]) : "v"(x[0]), "v"(b) : "vcc");
        } else if (KIND == 3) {                // v_readfirstlane
            int r;
            asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(r) : "v"(vint));
            asm volatile("s_add_u32 %0, %0, %1" : "+s"(sacc) : "s"(r));
        } else if (KIND == 4) {                // exec manipulation + branch over a vector instruction
            asm volatile("s_mov_b64 s[10:11], exec\n\ts_mov_b64 exec, 0xffff\n\tv_add_f32 %0, %0, %0\n\ts_mov_b64 exec, s[10:11]" : "+v"(x[2]) : : "s10", "s11");
        } else if (KIND == 5) {                // ds_write + ds_read with a wait
            lds[(wave & 3) * 1024 + lane] = x[3];
            x[3] += lds[(wave & 3) * 1024 + (lane ^ 1)];
        } else if (KIND == 6) {                // scalar branch on a vector compare (the usual `if (lane-dependent)` pattern)
            if (x[4] > 1e30f) x[5] += 1.f;
        } else if (KIND == 7) {                // v_readlane
            int r;
            asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(r) : "v"(vint));
            asm volatile("s_add_u32 %0, %0, %1" : "+s"(sacc) : "s"(r));
        } else if (KIND == 8) {                // v_writelane (an SGPR spill)
            asm volatile("v_writelane_b32 %0, %1, 5" : "+v"(vint) : "s"(sacc));
        } else if (KIND == 9) {                // global load + use
            x[6] += src[(lane + it * 64) & 4095];
        } else if (KIND == 10) {               // 64-bit integer multiply-add (address arithmetic)
            unsigned long long r = (unsigned long long)vint * 12345u + (unsigned)it;
            vint = (unsigned)(r >> 7);
        } else if (KIND == 11) {               // v_cmp -> SGPR pair used as a v_cndmask mask (no scalar reader)
            unsigned long long m;
            asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(x[0]), "v"(b));
            asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[7]) : "v"(b), "s"(m));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = (float)sacc + (float)vint;
    for (int i = 0; i < 16; ++i) s += x[i];
    dst[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NM, int KIND, int GAP>
static void run(const float *src, float *dst, long long *ticks, const char *what)
{
    const int nb = 256, iters = 400;
    double per[2];
    double mf = 0;
    for (int on = 0; on < 2; ++on) {
        (void)hipMemset(ticks, 0, nb * 8 * 8);
        hipLaunchKernelGGL((k<NM, KIND, GAP>), dim3(nb), dim3(512), 0, 0, src, dst, ticks, 8 * 4096 / NM, on, iters);
        (void)hipDeviceSynchronize();
        std::vector<long long> h(2 * nb * 8);
        (void)hipMemcpy(h.data(), ticks, 2 * nb * 8 * 8, hipMemcpyDeviceToHost);
        double s = 0, m = 0;
        for (int b = 0; b < nb; ++b) for (int w = 0; w < 4; ++w) { s += (double)h[b * 8 + 4 + w]; m += (double)h[b * 8 + w]; }
        if (on && KIND == 0) {
            // per probe wave: VGPR base of the wave and of the MFMA wave on its SIMD, and whether it was starved
            int hist[64][64][2] = {};
            for (int b = 0; b < nb; ++b)
                for (int w = 4; w < 8; ++w) {
                    const unsigned hw = (unsigned)h[2048 + b * 8 + w], ga = (unsigned)(h[2048 + b * 8 + w] >> 32);
                    int partner = -1;
                    for (int v = 0; v < 4; ++v) if ((((unsigned)h[2048 + b * 8 + v] >> 4) & 3) == ((hw >> 4) & 3)) partner = v;
                    const unsigned gp = partner >= 0 ? (unsigned)(h[2048 + b * 8 + partner] >> 32) : 0x3f;
                    const bool starved = (double)h[b * 8 + w] / iters > 1000.0;
                    hist[ga & 63][gp & 63][starved]++;
                }
            for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) if (hist[i][j][0] + hist[i][j][1])
                printf("      probe wave VGPR base %2d (x8 registers), MFMA wave of its SIMD base %2d: %4d ran, %4d starved\n", i, j, hist[i][j][0], hist[i][j][1]);
        }
        per[on] = s / (nb * 4) / iters;
        mf = m / (nb * 4) / (8.0 * 4096);
    }
    printf("MFMA blocks of %3d, %d x s_nop 15 behind each  probe: 16 fmas + %-44s cycles per iteration: alone %7.1f   beside the MFMA stream %7.1f   (%.1f cycles per MFMA)\n", NM, GAP, what, per[0], per[1], mf);
}

template <int KIND>
static void run3(const float *src, float *dst, long long *ticks, const char *what)
{
    run<8, KIND, 0>(src, dst, ticks, what);
    run<32, KIND, 0>(src, dst, ticks, what);
    run<128, KIND, 0>(src, dst, ticks, what);
    run<8, KIND, 3>(src, dst, ticks, what);
    run<32, KIND, 3>(src, dst, ticks, what);
    run<128, KIND, 3>(src, dst, ticks, what);
    run<32, KIND, 1>(src, dst, ticks, what);
    run<128, KIND, 1>(src, dst, ticks, what);
}

int main()
{
    float *src, *dst;
    long long *ticks;
    (void)hipMalloc(&src, 1 << 20);
    (void)hipMalloc(&dst, 256 * 512 * 4);
    (void)hipMalloc(&ticks, 2 * 256 * 8 * 8);
    (void)hipMemset(src, 0, 1 << 20);
    run3<0>(src, dst, ticks, "nothing else");
    run3<0>(src, dst, ticks, "nothing else");           // three times: which launches starve the probe is not a property of the code
    run3<0>(src, dst, ticks, "nothing else");
    run3<1>(src, dst, ticks, "v_cmp -> SGPR, scalar reader");
    run3<11>(src, dst, ticks, "v_cmp -> SGPR, v_cndmask reader");
    run3<2>(src, dst, ticks, "v_cmp -> VCC, v_cndmask reader");
    run3<3>(src, dst, ticks, "v_readfirstlane");
    run3<7>(src, dst, ticks, "v_readlane");
    run3<8>(src, dst, ticks, "v_writelane");
    run3<4>(src, dst, ticks, "exec change around a v_add");
    run3<6>(src, dst, ticks, "if (lane-dependent) { v_add }");
    run3<5>(src, dst, ticks, "ds_write, ds_read, wait, v_add");
    run3<9>(src, dst, ticks, "global load, wait, v_add");
    run3<10>(src, dst, ticks, "64-bit integer mad");
    return 0;
}
