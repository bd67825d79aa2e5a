// The K loop of a Winograd F(2x2, 3x3) convolution whose fp32 operands are split exactly into three bf16 pieces and multiplied on
// v_mfma_f32_32x32x16_bf16 -- as a bare loop, before the kernel is written: what does the instruction mix sustain on MI355X on random
// data (the 16-bit matrix cores are power-limited: tools/micro/mfma_f16_power.hip), and what do the transform + split instructions of
// the SAME wave cost beside the MFMAs?
//
// One 256-thread workgroup per CU (one wave per SIMD, 512 registers): wave i = plane row i of the 4 x 4 Winograd planes, a block of
// 8 x 8 tiles (two 32-tile MFMA blocks) x 64 output channels (two 32-channel blocks): 4 planes x 2 x 2 accumulators = 256 registers.
// Per 16-channel K step and wave: 32 ds_read_b128 of the raw patch (LDS image of the fp32 kernel, 16 channels per pixel), 128 adds
// (row combination + the four plane columns), 64 transformed values split into 3 bf16 pieces each (RNE: v_cvt_pk_bf16_f32, shift /
// mask, subtract: 5.5 instructions per value), 24 x 1 KB weight-fragment loads (3 pieces x 2 channel blocks x 4 planes, from L2),
// TERMS x 16 MFMAs (6 terms: 96 = 3072 matrix cycles).
//   MODE bits: 1 = transform + split (else the operand pieces are loop constants), 2 = weight fragments from global memory (else loop
//   constants), 4 = patch staging (6 buffer-like loads per thread per step, LDS commit, one barrier per step)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/wino_bf16_loop.hip -o /tmp/wino_bf16_loop && /tmp/wino_bf16_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int PIXQ = 5, ROWQ = 104, PH = 18, PW = 18;
constexpr int PBUF = PH * ROWQ + 8;       // quads per patch buffer (+ spare records for slots past the patch)
constexpr int NSLOT = PH * PW * 4, PS = (NSLOT + 255) / 256;

__device__ __forceinline__ unsigned cvt_pk(float a, float b)
{
    const bf16x2 v = __builtin_convertvector(f32x2{a, b}, bf16x2);      // v_cvt_pk_bf16_f32 (RNE)
    return __builtin_bit_cast(unsigned, v);
}
// x[0..7] -> three packed-bf16 fragments with x = h + m + l exactly (8 + 8 + 8 significand bits, round to nearest at each level)
__device__ __forceinline__ void split8(const float *x, u32x4 &ph, u32x4 &pm, u32x4 &pl)
{
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float x0 = x[2 * p], x1 = x[2 * p + 1];
        const unsigned h = cvt_pk(x0, x1);
        const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
        const unsigned m = cvt_pk(r0, r1);
        const float l0 = r0 - __builtin_bit_cast(float, m << 16), l1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
        ph[p] = h;
        pm[p] = m;
        pl[p] = cvt_pk(l0, l1);
    }
}

// the same pieces with the residuals taken by v_dot2c_f32_bf16 straight from the packed pair (x - h = x + h.lo * -1 + h.hi * 0): 3.5
// instructions per value instead of 5.5
__device__ __forceinline__ void split8_dot(const float *x, u32x4 &ph, u32x4 &pm, u32x4 &pl)
{
    const bf16x2 mlo = __builtin_bit_cast(bf16x2, 0x0000bf80u), mhi = __builtin_bit_cast(bf16x2, 0xbf800000u);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float x0 = x[2 * p], x1 = x[2 * p + 1];
        const unsigned h = cvt_pk(x0, x1);
        const float r0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, h), mlo, x0, false);
        const float r1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, h), mhi, x1, false);
        const unsigned m = cvt_pk(r0, r1);
        const float l0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, m), mlo, r0, false);
        const float l1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, m), mhi, r1, false);
        ph[p] = h;
        pm[p] = m;
        pl[p] = cvt_pk(l0, l1);
    }
}

// split check: h + m + l == x exactly, |m| <= 2^-8 |x|, |l| <= 2^-16 |x|, both methods the same pieces
__global__ void split_check(const float *x, unsigned *out, int n)
{
    const int i = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    u32x4 a[3], b[3];
    split8(x + i, a[0], a[1], a[2]);
    split8_dot(x + i, b[0], b[1], b[2]);
    for (int pc = 0; pc < 3; ++pc)
        for (int p = 0; p < 4; ++p) { out[(size_t)i * 3 + pc * 4 + p] = a[pc][p]; out[(size_t)n * 3 + (size_t)i * 3 + pc * 4 + p] = b[pc][p]; }
}

template <int TERMS, int MODE>
__global__ __launch_bounds__(256, 1) void k(const float *img, const u32x4 *wsrc, float *dst, unsigned long long *ticks, int nsteps)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4 *smem4 = reinterpret_cast<f32x4 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, g = lane >> 5;
    // fill the three patch buffers with this workgroup's (random) data
    const float *myimg = img + (size_t)blockIdx.x * (PH * PW * 128);
    for (int i = tid; i < 3 * PBUF; i += 256) {
        const float *s = myimg + (i * 4) % (PH * PW * 128 - 4);
        smem4[i] = f32x4{s[0], s[1], s[2], s[3]};
    }
    __syncthreads();
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sb = wave == 1 ? 1.f : -1.f;
    int abase[2], bbase[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int tyl = mb * 4 + (n >> 3), txl = n & 7;
        abase[mb] = (tyl + (ra >> 1) + (ra & 1) * (PH / 2)) * ROWQ + txl * PIXQ + 2 * g;
        bbase[mb] = (tyl + (rb >> 1) + (rb & 1) * (PH / 2)) * ROWQ + txl * PIXQ + 2 * g;
    }
    // staging slots (MODE & 4): slot = (pixel, quad) = (idx >> 2, idx & 3)
    unsigned pvo[PS];
    int plds[PS];
#pragma unroll
    for (int s = 0; s < PS; ++s) {
        const int idx = tid + 256 * s, pix = idx >> 2, py = pix / PW, px = pix - py * PW;
        pvo[s] = idx < NSLOT ? (unsigned)(pix * 128 + (idx & 3) * 4) : 0u;
        plds[s] = (idx < NSLOT ? ((py >> 1) + (py & 1) * (PH / 2)) * ROWQ + ((px >> 1) + (px & 1) * 9) * PIXQ : PH * ROWQ) + (idx & 3);
    }
    f32x16 acc[4][2][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][a][b][r] = 0.f;

    // t[mb][col][q]: row-combined patch columns of the current chunk (64 registers), V pieces [mb][piece] and weight pieces [nb][piece]
    f32x4 t[2][4][2];
    u32x4 V[2][2][3], U[2][2][3];
    const u32x4 *wbase = wsrc + (size_t)wave * (4 * 6 * 64) + lane;      // [step 8][wave 4][j 4][nb 2][piece 3][lane 64]
    int bo0 = 0, bo1 = PBUF, bo2 = 2 * PBUF;

#define TCOL(C, BOFF)                                                                                              \
    do {                                                                                                           \
        _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) _Pragma("unroll") for (int q = 0; q < 2; ++q) {           \
            const f32x4 a_ = smem4[(BOFF) + abase[mb] + (((C) >> 1) + ((C)&1) * 9) * PIXQ + q];                    \
            const f32x4 b_ = smem4[(BOFF) + bbase[mb] + (((C) >> 1) + ((C)&1) * 9) * PIXQ + q];                    \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) t[mb][C][q][e] = __builtin_fmaf(sb, b_[e], a_[e]);       \
        }                                                                                                          \
    } while (0)
// plane column J of both tile blocks from t, split into pieces -> V[SLOT]
#define VPLANE(J, SLOT)                                                                                            \
    do {                                                                                                           \
        _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) {                                                         \
            float v_[8];                                                                                           \
            _Pragma("unroll") for (int q = 0; q < 2; ++q) _Pragma("unroll") for (int e = 0; e < 4; ++e)            \
                v_[4 * q + e] = (J) == 0 ? t[mb][0][q][e] - t[mb][2][q][e]                                         \
                              : (J) == 1 ? t[mb][1][q][e] + t[mb][2][q][e]                                         \
                              : (J) == 2 ? t[mb][2][q][e] - t[mb][1][q][e] : t[mb][1][q][e] - t[mb][3][q][e];      \
            if ((MODE & 1) && !(MODE & 8)) split8(v_, V[SLOT][mb][0], V[SLOT][mb][1], V[SLOT][mb][2]);             \
            if ((MODE & 1) && (MODE & 8)) split8_dot(v_, V[SLOT][mb][0], V[SLOT][mb][1], V[SLOT][mb][2]);          \
        }                                                                                                          \
    } while (0)
#define ULOAD(STEP, J, SLOT)                                                                                       \
    do {                                                                                                           \
        if (MODE & 2) {                                                                                            \
            const u32x4 *w_ = wbase + ((size_t)((STEP)&7) * 16 + (J)) * (6 * 64);                                  \
            _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) _Pragma("unroll") for (int pc = 0; pc < 3; ++pc)      \
                U[SLOT][nb][pc] = w_[(nb * 3 + pc) * 64];                                                          \
        }                                                                                                          \
    } while (0)
#define ONE_MFMA(J, SLOT, WP, XP)                                                                                  \
    _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) _Pragma("unroll") for (int nb = 0; nb < 2; ++nb)              \
        acc[J][mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, U[SLOT][nb][WP]),      \
                                                                 __builtin_bit_cast(bf16x8, V[SLOT][mb][XP]), acc[J][mb][nb], 0, 0, 0)
// smallest terms first
#define MFMAS(J, SLOT)                                                                                             \
    do {                                                                                                           \
        if (TERMS >= 9) { ONE_MFMA(J, SLOT, 2, 2); }                                                               \
        if (TERMS >= 8) { ONE_MFMA(J, SLOT, 2, 1); ONE_MFMA(J, SLOT, 1, 2); }                                      \
        ONE_MFMA(J, SLOT, 2, 0); ONE_MFMA(J, SLOT, 1, 1); ONE_MFMA(J, SLOT, 0, 2);                                 \
        ONE_MFMA(J, SLOT, 1, 0); ONE_MFMA(J, SLOT, 0, 1); ONE_MFMA(J, SLOT, 0, 0);                                 \
    } while (0)

    // prologue: t of chunk 0, V of plane 0, U of plane 0
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                V[s][a][b] = wsrc[lane + 64 * (s * 6 + a * 3 + b)];
                U[s][a][b] = wsrc[lane + 64 * (12 + s * 6 + a * 3 + b)];
            }
    TCOL(0, bo0); TCOL(1, bo0); TCOL(2, bo0); TCOL(3, bo0);
    VPLANE(0, 0);
    ULOAD(0, 0, 0);
    const float *gimg = myimg;
    f32x4 pr[PS];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int c = 0; c < nsteps; ++c) {
        if (MODE & 4) {
#pragma unroll
            for (int s = 0; s < PS; ++s) pr[s] = *reinterpret_cast<const f32x4 *>(gimg + pvo[s] + ((c + 2) & 7) * 16);
        }
        // plane order 0, 3, 1, 2: t0 dies first, then t3, then t1 and t2 -- the next chunk's columns replace them as they die
        ULOAD(c, 3, 1);
        VPLANE(3, 1);
        MFMAS(0, 0);
        TCOL(0, bo1);
        ULOAD(c, 1, 0);
        VPLANE(1, 0);
        MFMAS(3, 1);
        TCOL(3, bo1);
        ULOAD(c, 2, 1);
        VPLANE(2, 1);
        MFMAS(1, 0);
        TCOL(1, bo1);
        TCOL(2, bo1);
        ULOAD(c + 1, 0, 0);
        VPLANE(0, 0);
        MFMAS(2, 1);
        if (MODE & 4) {
#pragma unroll
            for (int s = 0; s < PS; ++s) smem4[bo2 + plds[s]] = pr[s];
        }
        if (MODE & 16) {
#pragma unroll
            for (int i = 0; i < TERMS * 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (MODE & 8) ? 4 : 5, 0);
                if (i % 3 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (i % 3 == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        if (MODE & 4) __syncthreads();
        const int tt = bo0; bo0 = bo1; bo1 = bo2; bo2 = tt;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[j][a][b][r];
    if (!(MODE & 1)) s += t[0][0][0][0] + t[1][3][1][2];
    dst[blockIdx.x * 256 + tid] = s;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int TERMS, int MODE>
static void run(const char *tag, const float *img, const u32x4 *w, float *dst, unsigned long long *ticks, int nb, int nsteps)
{
    const int lds = 3 * PBUF * 16;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<TERMS, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<TERMS, MODE>), dim3(nb), dim3(256), lds, 0, img, w, dst, ticks, nsteps);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<TERMS, MODE>), dim3(nb), dim3(256), lds, 0, img, w, dst, ticks, nsteps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", tag); return; }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    std::vector<unsigned long long> c(nb);
    hipMemcpy(c.data(), ticks, nb * 8, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (auto v : c) cyc += (double)v;
    cyc /= nb;
    const double mf = (double)nb * 4 * nsteps * TERMS * 16.0;      // MFMAs per launch
    const double pf = mf * 32768.0 / (ms * 1e-3) / 1e15;
    // fp32 Winograd work this stands for: 16 (plane, block) products of 32 x 32 x 16 per step and wave
    const double tf32 = (double)nb * 4 * nsteps * 16.0 * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-46s %7.3f ms  %6.1f wave-cycles/MFMA  %5.3f PFLOP/s bf16  = %6.1f TFLOP/s of fp32 Winograd work (fp32 kernel: 112)  "
           "clock %.2f GHz\n", tag, ms, cyc / (nsteps * TERMS * 16.0), pf, tf32, cyc / (ms * 1e-3) / 1e9);
}

int main()
{
    const int nb = 256, nsteps = 4000;
    std::vector<float> himg((size_t)nb * PH * PW * 128);
    srand(3);
    for (auto &v : himg) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    std::vector<unsigned short> hw((size_t)8 * 16 * 6 * 64 * 8);
    for (auto &v : hw) {
        float f = (float)rand() / RAND_MAX * 2.f - 1.f;
        unsigned u;
        memcpy(&u, &f, 4);
        v = (unsigned short)(u >> 16);
    }
    float *img, *dst;
    u32x4 *w;
    unsigned long long *ticks;
    hipMalloc(&img, himg.size() * 4);
    hipMalloc(&w, hw.size() * 2);
    hipMalloc(&dst, nb * 256 * 4);
    hipMalloc(&ticks, nb * 8);
    hipMemcpy(img, himg.data(), himg.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    {   // the split itself: exact? both methods the same?
        const int n = 1 << 20;
        std::vector<float> hx(n);
        for (int i = 0; i < n; ++i) {
            const float mant = (float)rand() / RAND_MAX + 1.f;
            hx[i] = (rand() & 1 ? -1.f : 1.f) * ldexpf(mant, rand() % 60 - 30);
            if (i % 1000 == 0) hx[i] = 0.f;
            if (i % 1000 == 1) { unsigned u = 0x3f7fffffu + (i & 0x7fff0); memcpy(&hx[i], &u, 4); }
        }
        float *dx;
        unsigned *dout;
        hipMalloc(&dx, n * 4);
        hipMalloc(&dout, (size_t)n * 3 * 4 * 2);
        hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(split_check, dim3(n / 8 / 256), dim3(256), 0, 0, dx, dout, n);
        std::vector<unsigned> ho((size_t)n * 6);
        hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost);
        long bad_sum = 0, bad_same = 0;
        double worst_m = 0, worst_l = 0;
        auto bf = [](unsigned pk, int hi) { unsigned u = hi ? (pk & 0xffff0000u) : (pk << 16); float f; memcpy(&f, &u, 4); return (double)f; };
        for (int i = 0; i < n; i += 8)
            for (int p = 0; p < 4; ++p)
                for (int hi = 0; hi < 2; ++hi) {
                    const double x = hx[i + 2 * p + hi];
                    const unsigned *a = &ho[(size_t)i * 3], *b = &ho[(size_t)n * 3 + (size_t)i * 3];
                    const double h = bf(a[p], hi), m = bf(a[4 + p], hi), l = bf(a[8 + p], hi);
                    if (h + m + l != x) ++bad_sum;
                    if (a[p] != b[p] || a[4 + p] != b[4 + p] || a[8 + p] != b[8 + p]) ++bad_same;
                    if (x != 0) { worst_m = fmax(worst_m, fabs(m / x)); worst_l = fmax(worst_l, fabs(l / x)); }
                }
        printf("split check over %d values: h + m + l != x: %ld, dot2 pieces differ: %ld, max |m/x| = 2^%.2f, max |l/x| = 2^%.2f\n", n, bad_sum, bad_same,
               log2(worst_m), log2(worst_l));
    }
    for (int round = 0; round < 2; ++round) {
        run<6, 0>("6 terms, MFMAs alone (constant operands)", img, w, dst, ticks, nb, nsteps);
        run<6, 2>("6 terms, + weight fragments from L2", img, w, dst, ticks, nb, nsteps);
        run<6, 1>("6 terms, + transform and split", img, w, dst, ticks, nb, nsteps);
        run<6, 3>("6 terms, + both", img, w, dst, ticks, nb, nsteps);
        run<6, 7>("6 terms, + both + patch staging + barrier", img, w, dst, ticks, nb, nsteps);
        run<6, 15>("6 terms, everything, dot2 split", img, w, dst, ticks, nb, nsteps);
        run<6, 23>("6 terms, everything, interleave groups", img, w, dst, ticks, nb, nsteps);
        run<6, 31>("6 terms, everything, dot2 + groups", img, w, dst, ticks, nb, nsteps);
        run<8, 7>("8 terms, everything", img, w, dst, ticks, nb, nsteps);
        run<9, 7>("9 terms, everything", img, w, dst, ticks, nb, nsteps);
    }
    return 0;
}
