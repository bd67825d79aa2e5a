// Producer / consumer waves of one SIMD synchronised WITHOUT s_barrier: counters in LDS, polled.
// (tools/micro/ws_gap.hip: with one s_barrier per block of 32 MFMAs the second wave's vector work is not executed beside the MFMA
// stream but after it -- the step is the sum; free-running waves do overlap.)  Here wave i (0-3) streams blocks of 32
// v_mfma_f32_32x32x2_f32, wave 4 + i runs a producer-like step per block (8 ds_read_b128, 16 packed adds, 4 ds_write_b128, or 128
// fmas); they meet through two counters per pair: P[i] = steps the producer has finished, C[i] = blocks the consumer has finished.
// The consumer may start block s + 1 when P[i] >= s + 2 (it reads the counter one block early); the producer may start step s when
// C[i] >= s - RING + 1.  Spins are bounded: a stuck pair sets an error word instead of hanging the GPU.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ws_flag.hip -o tools/micro/ws_flag && tools/micro/ws_flag
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int PROBE, int RING, int SYNC>      // SYNC 0: free-running (no synchronisation), 1: LDS counters, 2: s_barrier per block
__global__ __launch_bounds__(512, 1) void k(const float *src, float *dst, long long *ticks, int blocks, int mfma_on, int *err)
{
    __shared__ f32x4 lds[2048];
    __shared__ volatile int cnt[64];       // [i] = C[i], [16 + i] = P[i]
    if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float a = src[lane], b = src[lane + 64];
    if (wave < 4) {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const long long t0 = __builtin_readcyclecounter();
        long long waited = 0;
        for (int s = 0; s < blocks; ++s) {
            int seen = 0;
            if (SYNC == 1) seen = cnt[16 + wave];          // read early, used after the block
            __builtin_amdgcn_sched_barrier(0);
            if (mfma_on) {
#pragma unroll
                for (int q = 0; q < 32; ++q) acc[q & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q & 7], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (SYNC == 1) {
                int p = __builtin_amdgcn_readfirstlane(seen);
                int spins = 0;
                while (p < s + 2 && s + 2 <= blocks) {          // the producer is normally ahead: no spin
                    __builtin_amdgcn_s_sleep(1);
                    p = __builtin_amdgcn_readfirstlane(cnt[16 + wave]);
                    if (++spins > 200000) { if (lane == 0) atomicAdd(err, 1); break; }
                }
                waited += spins;
                if (lane == 0) cnt[wave] = s + 1;
            }
            if (SYNC == 2) __syncthreads();
        }
        const long long t1 = __builtin_readcyclecounter();
        if (lane == 0) { ticks[blockIdx.x * 8 + wave] = t1 - t0; ticks[2048 + blockIdx.x * 8 + wave] = waited; }
        float sum = 0.f;
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
        dst[blockIdx.x * 512 + threadIdx.x] = sum;
        return;
    }
    const int pi = wave - 4;
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = a + (float)i;
    f32x4 q = {a, b, a, b};
    const long long t0 = __builtin_readcyclecounter();
    long long waited = 0;
    for (int s = 0; s < blocks; ++s) {
        if (SYNC == 1) {
            int spins = 0;
            while (__builtin_amdgcn_readfirstlane(cnt[pi]) < s - RING + 1) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > 200000) { if (lane == 0) atomicAdd(err, 1); break; }
            }
            waited += spins;
        }
        if (PROBE == 0) {
            for (int it = 0; it < 8; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = fmaf(x[i], 1.0001f, 0.5f);
            }
        } else {
            f32x4 r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = lds[1024 + pi * 64 + ((lane + i * 64) & 255)];
#pragma unroll
            for (int i = 0; i < 4; ++i) lds[pi * 256 + i * 64 + lane] = (r[i] + r[4 + i]) + q;
        }
        if (SYNC == 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the step's LDS writes are done before the counter says so
            if (lane == 0) cnt[16 + pi] = s + 1;
        }
        if (SYNC == 2) __syncthreads();
    }
    const long long t1 = __builtin_readcyclecounter();
    float sm = 0.f;
    for (int i = 0; i < 16; ++i) sm += x[i];
    dst[blockIdx.x * 512 + threadIdx.x] = sm + q[0];
    if (lane == 0) { ticks[blockIdx.x * 8 + wave] = t1 - t0; ticks[2048 + blockIdx.x * 8 + wave] = waited; }
}

template <int PROBE, int RING, int SYNC>
static void run(const float *src, float *dst, long long *ticks, int *err)
{
    const int nb = 256, blocks = 500;
    double per[2], mf[2], wc[2], wp[2];
    int herr[2] = {0, 0};
    for (int on = 0; on < 2; ++on) {
        (void)hipMemset(ticks, 0, 2 * nb * 8 * 8);
        (void)hipMemset(err, 0, 4);
        hipLaunchKernelGGL((k<PROBE, RING, SYNC>), dim3(nb), dim3(512), 0, 0, src, dst, ticks, blocks, on, err);
        (void)hipDeviceSynchronize();
        std::vector<long long> h(2 * nb * 8);
        (void)hipMemcpy(h.data(), ticks, 2 * nb * 8 * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&herr[on], err, 4, hipMemcpyDeviceToHost);
        double s = 0, m = 0, a = 0, c = 0;
        for (int b = 0; b < nb; ++b) for (int w = 0; w < 4; ++w) { s += (double)h[b * 8 + 4 + w]; m += (double)h[b * 8 + w]; a += (double)h[2048 + b * 8 + w]; c += (double)h[2048 + b * 8 + 4 + w]; }
        per[on] = s / (nb * 4) / blocks; mf[on] = m / (nb * 4) / blocks; wc[on] = a / (nb * 4) / blocks; wp[on] = c / (nb * 4) / blocks;
    }
    printf("%-14s ring %d  second wave: %-30s per block: second wave %7.1f alone, %7.1f beside MFMAs;  MFMA wave %7.1f (%6.1f without its MFMAs); spins per block: consumer %.2f, producer %.2f; stuck pairs %d %d\n",
           SYNC == 0 ? "free-running" : (SYNC == 1 ? "LDS counters" : "s_barrier"), RING, PROBE ? "8 ds_read + adds + 4 ds_write" : "128 fmas", per[0], per[1], mf[1], mf[0], wc[1], wp[1], herr[0], herr[1]);
    fflush(stdout);
}

int main()
{
    float *src, *dst;
    long long *ticks;
    int *err;
    (void)hipMalloc(&src, 1 << 20);
    (void)hipMalloc(&dst, 256 * 512 * 4);
    (void)hipMalloc(&ticks, 2 * 256 * 8 * 8);
    (void)hipMalloc(&err, 4);
    (void)hipMemset(src, 0, 1 << 20);
    for (int rep = 0; rep < 2; ++rep) {
        run<1, 2, 0>(src, dst, ticks, err);
        run<1, 2, 2>(src, dst, ticks, err);
        run<1, 2, 1>(src, dst, ticks, err);
        run<1, 3, 1>(src, dst, ticks, err);
        run<0, 2, 0>(src, dst, ticks, err);
        run<0, 2, 2>(src, dst, ticks, err);
        run<0, 2, 1>(src, dst, ticks, err);
        run<0, 3, 1>(src, dst, ticks, err);
    }
    return 0;
}
