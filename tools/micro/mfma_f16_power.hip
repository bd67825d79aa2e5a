// What does v_mfma_f32_32x32x16_f16 sustain on MI355X under the split-operand convolution's load?  (conv_split.hip issues six
// of them per product tile.)  Each wave loops: NR ds_read_b128 of operands from LDS, then 24 MFMAs on 4 accumulators -- the
// <2,2> wave tile of conv_split2_kernel: 12 reads per 24 MFMAs -- with operands that are random fp16 numbers, or all zeros
// (what the kernel's "no global loads" ablation computes on).  Prints FLOP/s and the shader clock the chip held
// (wave cycles from s_memtime... no: clock64() ticks of the shader clock / wall time of the launch).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f16_power.hip -o /tmp/mfma_f16_power && /tmp/mfma_f16_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// BAR: workgroup barriers per 3 iterations ("kernel row" of 3 taps): 0 none, 1, 2 (the shipped kernel's phase structure);
// WGS: workgroups per CU.  PKMUL: derive one operand per 6 MFMAs with v_pk_mul_f16 as conv_split.hip does.
template <int NR, int WAVES, int BAR = 0, int WGS = 1, bool PKMUL = false>
__global__ __launch_bounds__(WAVES * 64, WGS) void k(const _Float16 *src, float *dst, unsigned long long *clk, int iters)
{
    __shared__ __attribute__((aligned(16))) _Float16 lds[32768];
    const h8 k11 = {(_Float16)0x1p-11f, (_Float16)0x1p-11f, (_Float16)0x1p-11f, (_Float16)0x1p-11f,
                    (_Float16)0x1p-11f, (_Float16)0x1p-11f, (_Float16)0x1p-11f, (_Float16)0x1p-11f};
    for (int i = threadIdx.x; i < 32768; i += WAVES * 64) lds[i] = src[(blockIdx.x * 32768 + i) & 0xFFFFF];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const int base = ((it * 67 + lane + (threadIdx.x >> 6) * 19) * 8) & 16383;
        h8 f[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) f[j] = *reinterpret_cast<const h8 *>(lds + ((base + 1352 * (j % NR)) & 32760));
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if (PKMUL) f[6 + n * 3 + 2] = f[6 + n * 3] * k11;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[6 + n * 3 + q % 3], f[m * 3 + q / 2], acc[m * 2 + n], 0, 0, 0);
        }
        if (BAR && it % 3 == 2) {
            __syncthreads();
            if (BAR == 2) {
                lds[(threadIdx.x * 8 + it) & 32767] = (_Float16)acc[0][0];      // a token LDS write between the two barriers
                __syncthreads();
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    dst[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int NR, int WAVES, int BAR = 0, int WGS = 1, bool PKMUL = false>
static void run(const char *tag, const _Float16 *src, float *dst, unsigned long long *clk, int nb, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<NR, WAVES, BAR, WGS, PKMUL>), dim3(nb), dim3(WAVES * 64), 0, 0, src, dst, clk, iters);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<NR, WAVES, BAR, WGS, PKMUL>), dim3(nb), dim3(WAVES * 64), 0, 0, src, dst, clk, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(nb);
    hipMemcpy(c.data(), clk, nb * 8, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (auto v : c) cyc += (double)v;
    cyc /= nb;
    const double flop = (double)reps * nb * WAVES * iters * 24.0 * 32768.0;
    // one workgroup per CU, nb = 256: a launch lasts as long as one workgroup -> clock = cycles / time
    printf("%-34s %.3f ms per launch, %7.1f TFLOP/s f16 (%5.1f %% of 2516), wave cycles per MFMA %.1f, readcyclecounter rate %.2f GHz\n", tag,
           ms / reps, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 2.516e15 * 100, cyc / (iters * 24.0), cyc / (ms / reps * 1e-3) / 1e9);
}

int main()
{
    const int nb = 256, iters = 20000;
    std::vector<_Float16> h(1 << 20), z(1 << 20, (_Float16)0.f);
    srand(1);
    for (auto &v : h) v = (_Float16)((float)rand() / RAND_MAX * 2.f - 1.f);
    _Float16 *src, *zero;
    float *dst;
    unsigned long long *clk;
    hipMalloc(&src, h.size() * 2);
    hipMalloc(&zero, h.size() * 2);
    hipMalloc(&dst, 2 * nb * 512 * 4);
    hipMalloc(&clk, 2 * nb * 8);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(zero, z.data(), z.size() * 2, hipMemcpyHostToDevice);
    for (int round = 0; round < 2; ++round) {
        run<12, 8>("8 waves, 12 reads, random operands", src, dst, clk, nb, iters);
        run<12, 8>("8 waves, 12 reads, zero operands", zero, dst, clk, nb, iters);
        run<1, 8>("8 waves,  1 read,  random operands", src, dst, clk, nb, iters);
        run<12, 4>("4 waves, 12 reads, random operands", src, dst, clk, nb, iters);
        run<12, 4>("4 waves, 12 reads, zero operands", zero, dst, clk, nb, iters);
    }
    // the shipped kernel's skeleton: two 4-wave workgroups per CU (512 workgroups), barriers per kernel row
    run<12, 4, 0, 2>("2x4 waves, no barrier", src, dst, clk, 2 * nb, iters);
    run<12, 4, 1, 2>("2x4 waves, 1 barrier / 3 taps", src, dst, clk, 2 * nb, iters);
    run<12, 4, 2, 2>("2x4 waves, 2 barriers / 3 taps", src, dst, clk, 2 * nb, iters);
    run<12, 4, 2, 2, true>("2x4 waves, 2 barriers, pk_mul", src, dst, clk, 2 * nb, iters);
    run<12, 8, 1, 1>("1x8 waves, 1 barrier / 3 taps", src, dst, clk, nb, iters);
    return 0;
}
