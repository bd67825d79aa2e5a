// When does the second wave of a SIMD get vector instructions issued beside a wave that streams v_mfma_f32_32x32x2_f32?
// (tools/micro/ws_stall.hip: beside a gapless MFMA stream a vector probe keeps its pace; with one `s_nop 15` behind every MFMA it
// does not run at all.)  Here the MFMA wave's stream has the shape of a real K step: groups of 8 MFMAs with something between them.
// Waves 0-3: `blocks` x [4 groups of 8 MFMAs, BETWEEN after each group, END after the fourth]; waves 4-7: per block either 128 fmas
// (PROBE 0) or a producer-like step -- 8 ds_read_b128, 16 packed adds, 4 ds_write_b128 (PROBE 1).
//   BETWEEN: 0 nothing, 1 s_nop 0, 6 s_nop 3, 7 four scalar adds, 5 s_waitcnt 0 on nothing, 4 one v_add_u32, 2 two global loads + one
//            ds_read_b128 (results used 3 groups later), 3 = 2 + one v_add_u32
//   END:     0 nothing (free-running waves), 1 one s_barrier per block for both kinds of wave
//   BURST:   that many v_nop at the start of the second wave's step (does vector work in flight while the partner's first MFMA of the
//            block enters the pipe change anything?  no);  DELAY: s_nop 1 / 7 / 15 in the MFMA wave behind the barrier
// What it shows (round 5, profiles/r05_ws_gap.log): free-running, the producer-like step runs beside the MFMAs at its pace alone
// (229 vs 230 cycles) while the dense fma stream takes 5 x longer; with a barrier per block every mix is executed AFTER the block
// (the step is the sum), whatever BURST / DELAY.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ws_gap.hip -o tools/micro/ws_gap && tools/micro/ws_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int BETWEEN, int END, int PROBE, int BURST, int DELAY>
__global__ __launch_bounds__(512, 1) void k(const float *src, float *dst, long long *ticks, int blocks, int mfma_on)
{
    __shared__ f32x4 lds[2048];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float a = src[lane], b = src[lane + 64];
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(src);
    if (wave < 4) {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        f32x4 w0[4], w1[4], v[4];
        for (int i = 0; i < 4; ++i) { w0[i] = g4[lane + 64 * i]; w1[i] = g4[lane + 64 * i + 256]; v[i] = f32x4{a, b, a, b}; }
        unsigned vaddr = lane;
        const long long t0 = __builtin_readcyclecounter();
        for (int s = 0; s < blocks; ++s) {
            if (DELAY == 1) asm volatile("s_nop 1");
            if (DELAY == 2) asm volatile("s_nop 7");
            if (DELAY == 3) asm volatile("s_nop 15");
#pragma unroll
            for (int gp = 0; gp < 4; ++gp) {
                __builtin_amdgcn_sched_barrier(0);
                if (mfma_on) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[2 * gp] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[gp][j], v[gp][j], acc[2 * gp], 0, 0, 0);
                        acc[2 * gp + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[gp][j], v[gp][j], acc[2 * gp + 1], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (BETWEEN == 1) asm volatile("s_nop 0");
                if (BETWEEN == 6) asm volatile("s_nop 3");
                if (BETWEEN == 2 || BETWEEN == 3) {
                    w0[gp] = g4[(lane + 64 * gp + s * 256) & 4095];
                    w1[gp] = g4[(lane + 64 * gp + s * 256 + 2048) & 4095];
                    v[gp] = lds[(vaddr & 1023) + 256 * (gp & 3)];
                }
                if (BETWEEN == 3 || BETWEEN == 4) asm volatile("v_add_u32 %0, %0, 1" : "+v"(vaddr));
                if (BETWEEN == 5) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
                if (BETWEEN == 7) asm volatile("s_add_u32 s20, s20, 1\n\ts_add_u32 s20, s20, 1\n\ts_add_u32 s20, s20, 1\n\ts_add_u32 s20, s20, 1" : : : "s20");
            }
            __builtin_amdgcn_sched_barrier(0);
            if (END == 1) __syncthreads();
        }
        const long long t1 = __builtin_readcyclecounter();
        if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
        float sum = (float)vaddr;
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
        dst[blockIdx.x * 512 + threadIdx.x] = sum;
        return;
    }
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = a + (float)i;
    for (int i = 0; i < 3000; ++i) a = a * 1.0001f + 0.25f;      // let the MFMA stream start
    const long long t0 = __builtin_readcyclecounter();
    f32x4 q = {a, b, a, b};
    for (int s = 0; s < blocks; ++s) {
#pragma unroll
        for (int i = 0; i < BURST; ++i) asm volatile("v_nop");        // vector instructions in flight while the partner's first MFMA of the block enters the pipe
        if (PROBE == 0) {
            for (int it = 0; it < 8; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = fmaf(x[i], 1.0001f, 0.5f);
            }
        } else {   // a producer-like step: 8 ds_read_b128, 32 adds, 4 ds_write_b128
            f32x4 r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = lds[1024 + (wave & 3) * 64 + ((lane + i * 64) & 255)];
#pragma unroll
            for (int i = 0; i < 4; ++i) lds[1024 + 256 + (wave & 3) * 256 + i * 64 + lane] = (r[i] + r[4 + i]) + q;
        }
        if (END == 1) __syncthreads();
    }
    const long long t1 = __builtin_readcyclecounter();
    float sm = 0.f;
    for (int i = 0; i < 16; ++i) sm += x[i];
    dst[blockIdx.x * 512 + threadIdx.x] = sm + q[0];
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int BETWEEN, int END, int PROBE, int BURST, int DELAY>
static void run(const float *src, float *dst, long long *ticks, const char *what)
{
    const int nb = 256, blocks = 1000;
    double per[2], mf[2];
    for (int on = 0; on < 2; ++on) {
        (void)hipMemset(ticks, 0, nb * 8 * 8);
        hipLaunchKernelGGL((k<BETWEEN, END, PROBE, BURST, DELAY>), dim3(nb), dim3(512), 0, 0, src, dst, ticks, blocks, on);
        (void)hipDeviceSynchronize();
        std::vector<long long> h(nb * 8);
        (void)hipMemcpy(h.data(), ticks, nb * 8 * 8, hipMemcpyDeviceToHost);
        double s = 0, m = 0;
        for (int b = 0; b < nb; ++b) for (int w = 0; w < 4; ++w) { s += (double)h[b * 8 + 4 + w]; m += (double)h[b * 8 + w]; }
        per[on] = s / (nb * 4) / blocks;
        mf[on] = m / (nb * 4) / blocks;
    }
    printf("%-62s %-9s burst %2d delay %d %s: second wave per block: %7.1f without MFMAs, %7.1f beside them;  MFMA wave per block of 32: %7.1f (%.1f without its MFMAs)\n", what,
           END == 1 ? "barrier" : "free", BURST, DELAY, PROBE ? "8 ds_read + adds + 4 ds_write" : "128 fmas", per[0], per[1], mf[1], mf[0]);
}

template <int BETWEEN>
static void run4(const float *src, float *dst, long long *ticks, const char *what)
{
    run<BETWEEN, 0, 1, 0, 0>(src, dst, ticks, what);
    run<BETWEEN, 1, 1, 0, 0>(src, dst, ticks, what);
    run<BETWEEN, 1, 1, 8, 0>(src, dst, ticks, what);
    run<BETWEEN, 1, 1, 8, 1>(src, dst, ticks, what);
    run<BETWEEN, 1, 1, 24, 1>(src, dst, ticks, what);
    run<BETWEEN, 1, 1, 24, 2>(src, dst, ticks, what);
    run<BETWEEN, 1, 1, 24, 3>(src, dst, ticks, what);
    run<BETWEEN, 1, 1, 48, 3>(src, dst, ticks, what);
    run<BETWEEN, 1, 0, 24, 2>(src, dst, ticks, what);
}

int main()
{
    float *src, *dst;
    long long *ticks;
    (void)hipMalloc(&src, 1 << 20);
    (void)hipMalloc(&dst, 256 * 512 * 4);
    (void)hipMalloc(&ticks, 256 * 8 * 8);
    (void)hipMemset(src, 0, 1 << 20);
    for (int rep = 0; rep < 1; ++rep) {
        run4<0>(src, dst, ticks, "between groups of 8 MFMAs: nothing");
        run4<2>(src, dst, ticks, "between groups of 8 MFMAs: 2 global loads + ds_read_b128");
        run4<3>(src, dst, ticks, "between groups of 8 MFMAs: 2 global loads + ds_read + v_add");
    }
    return 0;
}
