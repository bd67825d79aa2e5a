// Flow-head convolution: 32 -> 2 channels, k x k (k = 3, 5, 7), stride 1, bias + residual, no activation
// (reference: the last torch.nn.Conv2d of conv_M / conv_S, /root/reference/src/models.py:161-162, 205-206, and the
// "+ xflow" of :186 / :216).  On the matrix cores this layer wastes 30 of 32 output columns, so it runs on the VALU:
//   * one workgroup = 16 x 16 output pixels, one pixel per lane; the (16+k-1)^2 x 32-channel input patch is staged
//     once in LDS as 128-byte pixel vectors whose 16-byte quads are XOR-swizzled by (column>>1)&7, which makes every
//     ds_read_b128 of a 16-pixel row segment conflict-free for every tap;
//   * the 2 x 32 x k x k weights are read through the scalar cache (wave-uniform addresses -> s_load), so each FMA
//     takes its weight from an SGPR and one 16-byte LDS read feeds 8 FMAs (balanced LDS : VALU on CDNA4);
//   * output is the network's 4-lane flow layout (u, v, 0, 0).
#include <algorithm>
#include "common.h"

namespace pivlfn {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

// (a0, a1) += x * (w0, w1): one v_pk_fma_f32 -- both outputs of a pixel per instruction, the weight pair from one 64-bit scalar
// operand.  Per output the chain of fused multiply-adds is the one of the scalar form, so the bits do not change.
__device__ __forceinline__ f32x2 fma2(float x, const float *w2, f32x2 a)
{
    return __builtin_elementwise_fma(f32x2{x, x}, *reinterpret_cast<const f32x2 *>(w2), a);
}

// WLDS: weights staged in LDS and read as broadcast ds_reads instead of through the scalar cache.  On a grid of a few
// workgroups (the coarse levels) the K*K*8 dependent s_load round trips are the whole kernel (22 us for 32x32 pixels); with many
// waves per CU they hide behind each other and the scalar path is the better one (it keeps LDS bandwidth for the patch).
template <int K, bool WLDS>
__global__ __launch_bounds__(256) void conv_head_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                        float b0, float b1, const float *__restrict__ res4,
                                                        float *__restrict__ out4, int B, int H, int W)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int P = K / 2, PW = 16 + K - 1, NPIX = PW * PW;
    const int tiles_x = (W + 15) >> 4, tiles_y = (H + 15) >> 4;
    int bid = xcd_remap(blockIdx.x, tiles_x * tiles_y * B);
    const int tx0 = (bid % tiles_x) * 16;
    bid /= tiles_x;
    const int ty0 = (bid % tiles_y) * 16;
    const int b = bid / tiles_y;
    const int tid = threadIdx.x;
    const float *xb = x + (size_t)b * H * W * 32;
    float *wl = smem + NPIX * 32;
    if (WLDS)
        for (int i = tid; i < K * K * 64; i += 256) wl[i] = w[i];

    for (int idx = tid; idx < NPIX * 8; idx += 256) {
        const int pix = idx >> 3, q = idx & 7;
        const int r = pix / PW, c = pix - r * PW;
        const int iy = ty0 + r - P, ix = tx0 + c - P;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const f32x4 *>(xb + ((size_t)iy * W + ix) * 32 + 4 * q);
        *reinterpret_cast<f32x4 *>(smem + pix * 32 + 4 * (q ^ ((c >> 1) & 7))) = v;
    }
    __syncthreads();

    const int lx = tid & 15, ly = tid >> 4;
    const char *sb = reinterpret_cast<const char *>(smem);
    f32x2 a = {0.f, 0.f};
#pragma unroll 1
    for (int ky = 0; ky < K; ++ky) {
        const float *wr = (WLDS ? wl : w) + ky * K * 64;      // [kx][quad][4][out]
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const int c = lx + kx;
            const unsigned base = (unsigned)((ly + ky) * PW + c) * 128u | (unsigned)(((c >> 1) & 7) << 4);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(sb + (base ^ (unsigned)(16 * q)));
                const float *wq = wr + (kx * 8 + q) * 8;
                a = fma2(v[0], wq, a); a = fma2(v[1], wq + 2, a); a = fma2(v[2], wq + 4, a); a = fma2(v[3], wq + 6, a);
            }
        }
    }
    const int oy = ty0 + ly, ox = tx0 + lx;
    if (oy < H && ox < W) {
        const size_t pix = ((size_t)b * H + oy) * W + ox;
        f32x4 o = {a[0] + b0, a[1] + b1, 0.f, 0.f};
        if (res4) {
            const float2 r = *reinterpret_cast<const float2 *>(res4 + pix * 4);
            o[0] += r.x;
            o[1] += r.y;
        }
        *reinterpret_cast<f32x4 *>(out4 + pix * 4) = o;
    }
}

// ---- large images (levels 1 and 2): four output rows per lane ---------------------------------------------------------
// The kernel above reads one 16-byte quad of the patch per 8 fused multiply-adds and is bound by LDS read bandwidth and the
// latency of those reads at two workgroups per CU (a 7x7 head at level 1: 232 us, 20 % of the VALU rate).  Here a lane owns a
// column of FOUR vertically adjacent output pixels: a quad read at patch row r serves the taps ky = r - i of all four rows i, so
// ten reads feed 28 (row, tap) products -- 2.8x less LDS traffic per pixel -- and a weight taken from a scalar register is used by
// four pixels.  Lanes are consecutive pixels in x (conflict-free 16-byte reads with the quad pair of a pixel swapped for
// columns 8..15 of every 16); a wave covers 16 x 16 pixels, a workgroup 32 x 32; the 32 input channels are staged in four
// passes of 8 (a 38 x 38 x 32-byte patch, 46 KB: three workgroups per CU).
template <int K>
__global__ __launch_bounds__(256, 3) void conv_head4_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                            float b0, float b1, const float *__restrict__ res4,
                                                            float *__restrict__ out4, int B, int H, int W)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int P = K / 2, PW = 32 + K - 1, NPIX = PW * PW, NR = 4 + K - 1;
    const int tiles_x = (W + 31) >> 5, tiles_y = (H + 31) >> 5;
    int bid = xcd_remap(blockIdx.x, tiles_x * tiles_y * B);
    const int tx0 = (bid % tiles_x) * 32;
    bid /= tiles_x;
    const int ty0 = (bid % tiles_y) * 32;
    const int b = bid / tiles_y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *xb = x + (size_t)b * H * W * 32;
    const int lx = (wave & 1) * 16 + (lane & 15);          // column of this lane inside the tile
    const int ly0 = (wave >> 1) * 16 + (lane >> 4) * 4;    // first of its four rows

    f32x2 a[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = f32x2{0.f, 0.f};

#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
        if (pass) __syncthreads();
        for (int idx = tid; idx < NPIX * 2; idx += 256) {
            const int pix = idx >> 1, qq = idx & 1;
            const int r = pix / PW, c = pix - r * PW;
            const int iy = ty0 + r - P, ix = tx0 + c - P;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const f32x4 *>(xb + ((size_t)iy * W + ix) * 32 + pass * 8 + qq * 4);
            *reinterpret_cast<f32x4 *>(smem + pix * 8 + 4 * (qq ^ ((c >> 3) & 1))) = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int kx = 0; kx < K; ++kx) {
            const int c = lx + kx;
            const int sw = (c >> 3) & 1;
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const float *wq = w + (kx * 8 + pass * 2 + qq) * 8;          // + ky * K * 64: [tap][quad][4][out]
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(smem + ((ly0 + r) * PW + c) * 8 + 4 * (qq ^ sw));
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int ky = r - i;
                        if (ky >= 0 && ky < K) {
                            const float *wk = wq + ky * K * 64;
                            a[i] = fma2(v[0], wk, a[i]); a[i] = fma2(v[1], wk + 2, a[i]);
                            a[i] = fma2(v[2], wk + 4, a[i]); a[i] = fma2(v[3], wk + 6, a[i]);
                        }
                    }
                }
            }
        }
    }
    const int ox = tx0 + lx;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int oy = ty0 + ly0 + i;
        if (oy < H && ox < W) {
            const size_t pix = ((size_t)b * H + oy) * W + ox;
            f32x4 o = {a[i][0] + b0, a[i][1] + b1, 0.f, 0.f};
            if (res4) {
                const float2 r = *reinterpret_cast<const float2 *>(res4 + pix * 4);
                o[0] += r.x;
                o[1] += r.y;
            }
            *reinterpret_cast<f32x4 *>(out4 + pix * 4) = o;
        }
    }
}

// ---- large images, wide kernels: the head as a (K x 1) convolution on the fp32 matrix cores + a diagonal sum -----------------
// out[y][x][o] = b[o] + sum_kx T[y][x + kx - p][o, kx],   T[y][x'][o, kx] = sum_{ky, c} in[y + ky - p][x'][c] * w[o][c][ky][kx].
// T is a vertical (K x 1) convolution from 32 to 2 x K <= 16 channels: on v_mfma_f32_16x16x4_f32 (M = the 16 (o, kx) slots, N = 16
// pixels of a row, K = 4 channels) it wastes 2 of 16 output rows at K = 7 instead of the 30 of 32 columns the plain head would, and
// needs 3.5 MFMAs per pixel: 48 us of matrix time at 1024^2 where the vector kernel above is LDS-bound at 150 us.
//   * workgroup = 4 waves = a strip of 64 columns (58 outputs + the 6-column halo of the diagonal sum) x 16 rows; wave w owns
//     columns 16 w .. 16 w + 15 of all rows.  No LDS and no barrier in the matrix phase: the B operand (16 pixels x 16 channels per
//     16-byte load) comes straight from global memory through a buffer descriptor (rows and columns outside the image read as
//     zeros: the convolution's padding), every input row is loaded once and feeds the K output rows it belongs to; the A operand
//     (weights, 2 K fragments packed at load time) stays in registers;
//   * the T tile goes through LDS (8 rows x 64 columns x 16 slots at a time) for the diagonal sum, bias and residual; output in the
//     4-lane flow layout.
using f32x4h = __attribute__((ext_vector_type(4))) float;
using u32x4h = __attribute__((ext_vector_type(4))) unsigned int;
template <int K, int TH>
__global__ __launch_bounds__(256) void conv_head_mfma_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                             float b0, float b1, const float *__restrict__ res4,
                                                             float *__restrict__ out4, int B, int H, int W)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int P = K / 2, TWO = 64 - 2 * P;
    const int tiles_x = (W + TWO - 1) / TWO, tiles_y = (H + TH - 1) / TH;
    int bid = xcd_remap(blockIdx.x, tiles_x * tiles_y * B);
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int xs0 = tx * TWO - P, y0 = ty * TH;          // first column of the strip (may be negative), first output row
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kq = lane >> 4;

    // weight fragments, packed at load time behind the vector kernels' table (pack_head): one 16-byte load per (ky, half)
    f32x4h A[K][2];
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int h = 0; h < 2; ++h) A[ky][h] = *reinterpret_cast<const f32x4h *>(w + K * K * 64 + ((ky * 2 + h) * 64 + lane) * 4);

    const size_t img = (size_t)H * W;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x + (size_t)b * img * 32), 0, (unsigned)(img * 32 * sizeof(float)), 0x00020000);
    const int xs = xs0 + 16 * wave + n16;              // this lane's column
    const bool xin = xs >= 0 && xs < W;
    f32x4h acc[TH];
#pragma unroll
    for (int i = 0; i < TH; ++i) acc[i] = f32x4h{0.f, 0.f, 0.f, 0.f};
    // input rows are loaded three rows ahead of their MFMAs (a row's 2 x 16 bytes per lane come from HBM: ~1-2 us under load, a
    // row's matrix work is 0.75 us); the fence keeps the compiler from sinking the loads back to their use
    constexpr int NR = TH + 2 * P, AHEAD = 3;
    f32x4h Bq[NR][2];
#define HEAD_LOAD(RI)                                                                             \
    do {                                                                                          \
        const int r_ = y0 - P + (RI);                                                             \
        const unsigned off_ = (xin && r_ >= 0 && r_ < H) ? (unsigned)((r_ * W + xs) * 32 + 4 * kq) * 4u : 0x80000000u; \
        Bq[RI][0] = __builtin_bit_cast(f32x4h, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off_, 0, 0)); \
        Bq[RI][1] = __builtin_bit_cast(f32x4h, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off_, 64, 0)); \
    } while (0)
#pragma unroll
    for (int ri = 0; ri < AHEAD; ++ri) HEAD_LOAD(ri);
#pragma unroll
    for (int ri = 0; ri < NR; ++ri) {
        if (ri + AHEAD < NR) HEAD_LOAD(ri + AHEAD);
        __builtin_amdgcn_sched_barrier(0);
        // tap ky of input row ri goes to output row ri - ky: consecutive MFMAs target different accumulators (the instruction's
        // dependent latency is 40 cycles against 32 of issue); per accumulator the order stays (row, half, j)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ky = 0; ky < K; ++ky) {
                    const int yi = ri - ky;
                    if (yi < 0 || yi >= TH) continue;
                    acc[yi] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[ky][h][j], Bq[ri][h][j], acc[yi], 0, 0, 0);
                }
    }
#undef HEAD_LOAD
    // T tile -> LDS in two passes of 8 rows (32 KB: four workgroups per CU): [row][column 64][16 slots]; lane holds slots
    // 4 kq .. 4 kq + 3 of column 16 wave + n16
    f32x4h *T4 = reinterpret_cast<f32x4h *>(smem);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
#pragma unroll
        for (int i = 0; i < TH / 2; ++i) T4[(i * 64 + 16 * wave + n16) * 4 + kq] = acc[half * (TH / 2) + i];
        __syncthreads();
        for (int idx = tid; idx < (TH / 2) * TWO; idx += 256) {
            const int yi = idx / TWO, xo = idx - yi * TWO;
            const int ox = tx * TWO + xo, oy = y0 + half * (TH / 2) + yi;
            if (ox >= W || oy >= H) continue;
            const float *t = smem + (yi * 64 + xo) * 16;
            float u = b0, v = b1;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                u += t[kx * 16 + kx];
                v += t[kx * 16 + 8 + kx];
            }
            const size_t pix = (size_t)(b * H + oy) * W + ox;
            if (res4) {
                const f32x4h rr = *reinterpret_cast<const f32x4h *>(res4 + pix * 4);
                u += rr[0];
                v += rr[1];
            }
            *reinterpret_cast<f32x4h *>(out4 + pix * 4) = f32x4h{u, v, 0.f, 0.f};
        }
    }
}

// ---- the same structure for Regularization's (7 x 1) distance convolution, 32 -> 49 channels (conv_dist_R.0 of levels 1 and 2,
// /root/reference/src/models.py:253-257): four blocks of 16 output channels, one per wave, 16 columns x TH rows per workgroup.  The
// direct kernel pads 49 channels to two 32-wide blocks and stages a 22-row patch through LDS per 8 input channels; here every
// wave streams its 16 columns' rows once from global memory (the four waves of a workgroup read the same bytes: L1 hits) against
// 14 register-resident weight fragments, with no LDS and no barrier, and stores 16-byte channel quads straight from the
// accumulator layout.  No activation (the reference has none between the two halves of the separable pair).
template <int TH>
__global__ __launch_bounds__(256, 2) void conv_col7_kernel(const float *__restrict__ x, int x_stride, const float *__restrict__ wf,
                                                        const float *__restrict__ bias, float *__restrict__ out, int out_stride,
                                                        int cout_store, int single48, int B, int H, int W)
{
    constexpr int K = 7, P = 3, NR = TH + 2 * P, AHEAD = 3;
    const int tiles_x = (W + 15) >> 4, tiles_y = (H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * B;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kq = lane >> 4;
    // A workgroup takes four consecutive tiles, one after the other, and wave w takes output block (w + i) & 3 of the i-th: in the
    // 49-channel layer block 3 is channel 48 alone, on the vector unit, a fraction of the matrix work of the other blocks.  With one
    // tile per workgroup the wave that drew it left its slot empty until the workgroup retired.  The waves share no LDS and meet at
    // no barrier: every one runs three matrix blocks and one vector block, in its own order.  (Persistent workgroups, two per CU,
    // walking the tiles pass by pass: no better -- 222 / 330 us against 225 / 312 at 1024 x 1024.)
#pragma unroll 1
    for (int ph = 0; ph < 4; ++ph) {
    int bid = xcd_remap(blockIdx.x, gridDim.x) * 4 + ph;
    if (bid >= ntiles) break;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int y0 = ty * TH;
    const int blk = (wave + ph) & 3;                                 // 16-channel output block of this wave in this tile
    f32x4h A[K][2];
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int h = 0; h < 2; ++h) A[ky][h] = *reinterpret_cast<const f32x4h *>(wf + (((blk * K + ky) * 2 + h) * 64 + lane) * 4);
    const size_t img = (size_t)H * W;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x + (size_t)b * img * x_stride), 0,
                                                                          (unsigned)(((img - 1) * x_stride + 32) * sizeof(float)), 0x00020000);
    const int xs = tx * 16 + n16;
    const bool xin = xs < W;
    // lane holds output channels 16 blk + 4 kq .. + 3 of column xs for every row.  The accumulators start from the bias, and a row is
    // stored as soon as its seventh tap is in (row ri - 6 after input row ri): one 16-byte buffer store straight from the accumulator,
    // under the next rows' MFMAs.  (An epilogue of `store(acc[i] + bias)` put every sum into one temporary, and the wait protecting a
    // store's source registers made each of the 16 stores wait for the one before it: a third of the workgroup's life.)
    const int ch = 16 * blk + 4 * kq;
    const bool vec48 = blk == 3 && single48;
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)b * img * out_stride, 0,
                                                                           (unsigned)(((img - 1) * out_stride + cout_store) * sizeof(float)), 0x00020000);
    const bool chin = xin && ch < cout_store;
    f32x4h acc[TH];           // acc[i] is live from input row i (its first tap) to input row i + 6 (its store): seven or eight at a time
    f32x4h b4 = *reinterpret_cast<const f32x4h *>(bias + ch);
    if (vec48 && kq != 0) b4 = f32x4h{0.f, 0.f, 0.f, 0.f};      // the vector path adds its four lane groups at the end: the bias once
    f32x4h Bq[NR][2];
#define COL_LOAD(RI)                                                                              \
    do {                                                                                          \
        const int r_ = y0 - P + (RI);                                                             \
        const unsigned off_ = (xin && r_ >= 0 && r_ < H) ? (unsigned)((r_ * W + xs) * x_stride + 4 * kq) * 4u : 0x80000000u; \
        Bq[RI][0] = __builtin_bit_cast(f32x4h, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off_, 0, 0)); \
        Bq[RI][1] = __builtin_bit_cast(f32x4h, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off_, 64, 0)); \
    } while (0)
#pragma unroll
    for (int ri = 0; ri < AHEAD; ++ri) COL_LOAD(ri);
#pragma unroll
    for (int ri = 0; ri < NR; ++ri) {
        if (ri + AHEAD < NR) COL_LOAD(ri + AHEAD);
        if (ri < TH) acc[ri] = b4;
        __builtin_amdgcn_sched_barrier(0);
        if (vec48) {
            // 49 output channels = three 16-channel blocks + ONE channel: a fourth MFMA block would spend 15 of its 16 rows on
            // padding.  The fourth wave takes channel 48 on the vector unit instead: its fragment registers hold that channel's
            // weights (replicated over the slots at pack time), every lane multiplies its 8 channels of the row, the four
            // channel groups of a pixel are added across lanes at the end.  56 fused multiply-adds per input row where the
            // matrix version issues 56 MFMAs of 32 cycles.
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int yi = ri - ky;
                if (yi < 0 || yi >= TH) continue;
                float e = acc[yi][0];
#pragma unroll
                for (int j = 0; j < 4; ++j) e = fmaf(A[ky][0][j], Bq[ri][0][j], e);
#pragma unroll
                for (int j = 0; j < 4; ++j) e = fmaf(A[ky][1][j], Bq[ri][1][j], e);
                acc[yi][0] = e;
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ky = 0; ky < K; ++ky) {       // consecutive MFMAs target different accumulators; per accumulator: (row, half, j)
                        const int yi = ri - ky;
                        if (yi < 0 || yi >= TH) continue;
                        acc[yi] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[ky][h][j], Bq[ri][h][j], acc[yi], 0, 0, 0);
                    }
        }
        if (ri >= K - 1) {                     // output row ri - 6 is complete
            const int yi = ri - (K - 1);
            if (vec48) {                       // channel groups kq = 0..3 of a pixel live in lanes l, l+16, l+32, l+48: add them (fixed order)
                float e = acc[yi][0];
                e = e + __shfl_xor(e, 16);
                e = e + __shfl_xor(e, 32);
                acc[yi] = f32x4h{e, 0.f, 0.f, 0.f};
            }
            const int oy = y0 + yi;
            const unsigned off = (chin && oy < H) ? (unsigned)((oy * W + xs) * out_stride + ch) * 4u : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4h, acc[yi]), rso, (int)off, 0, 0);
        }
    }
#undef COL_LOAD
    }       // tiles of the workgroup
}

// wf: [4 blocks][7][2][64][4] fragments (net.hip pack_conv), bias: [64]; single48: the layer has 49 output channels and block 3 of
// wf holds channel 48's weights replicated over the slots (vector path of the fourth wave)
int launch_conv_col7(const float *x, int x_stride, const float *wf, const float *bias, float *out, int out_stride, int cout_store,
                     int single48, int B, int H, int W, hipStream_t st)
{
    PIV_REQUIRE(x && wf && bias && out && B > 0 && H > 0 && W > 0, "conv_col7: bad arguments");
    PIV_REQUIRE(x_stride % 4 == 0 && out_stride % 4 == 0 && cout_store % 4 == 0 && cout_store <= 64 && cout_store <= out_stride, "conv_col7: bad strides");
    // the 49-channel path leaves channel 48's value in all four lane groups of block 3 and relies on only lane group 0 being stored
    PIV_REQUIRE(!single48 || cout_store == 52, "conv_col7: the 49-channel layer stores 52 lanes (got %d)", cout_store);
    PIV_REQUIRE((long)H * W * std::max(x_stride, out_stride) * 4 < (1L << 31), "conv_col7: image exceeds 2 GiB (32-bit buffer offsets for the loads and the stores)");
    constexpr int TH = 16;
    hipLaunchKernelGGL((conv_col7_kernel<TH>), dim3(cdiv(cdiv(W, 16) * cdiv(H, TH) * B, 4)), dim3(256), 0, st, x, x_stride, wf, bias, out, out_stride, cout_store, single48, B, H, W);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

// ---- (1 x 7) distance convolution, 49 -> 49 channels on 52 stored lanes (conv_dist_R.1 of levels 1 and 2,
// /root/reference/src/models.py:258-261): conv_col7 turned by 90 degrees.  The 16 "pixels" of the matrix instruction are 16 ROWS of
// one image column; a wave walks along x and every column it loads (13 channel quads: three 16-byte loads of quads kq, kq+4, kq+8
// and channel 48 on its own) feeds the seven outputs x - 3 .. x + 3 from registers: the shift of a tap is a different accumulator,
// never a different lane.  7 x 13 = 91 MFMAs per column and 16-channel block (the direct kernel's 64 x 56 padding: 7 x 14 x 2 of
// twice the size), no LDS, no barrier; the four waves of a workgroup read the same bytes (L1 hits) and own one output block each,
// block 3 -- channel 48 alone -- on the vector unit.  No activation, bias added.
template <int TW>
__global__ __launch_bounds__(256, 2) void conv_row7_kernel(const float *__restrict__ x, int x_stride, const float *__restrict__ wf,
                                                        const float *__restrict__ wf12, const float *__restrict__ bias,
                                                        float *__restrict__ out, int out_stride, int B, int H, int W,
                                                        unsigned long long *stamps)
{
    constexpr int K = 7, P = 3, NC = TW + 2 * P, AHEAD = 3;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + 15) >> 4;
    const int ntiles = tiles_x * tiles_y * B;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kq = lane >> 4;
    // four consecutive tiles per workgroup, wave w on block (w + i) & 3 of the i-th (see conv_col7_kernel)
#pragma unroll 1
    for (int ph = 0; ph < 4; ++ph) {
    int bid = xcd_remap(blockIdx.x, gridDim.x) * 4 + ph;
    if (bid >= ntiles) break;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int x0 = tx * TW;
    const int blk = (wave + ph) & 3;
#ifdef PIVLFN_STAMPS          // tools build: start of the unit, weights there, first column there, last MFMA issued, end (s_memtime ticks)
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
#define R7_NOW(T) do { __builtin_amdgcn_sched_barrier(0); T = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    if (stamps) R7_NOW(ts0);
#endif
    f32x4h A[K][3];
    float A12[K];
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
#pragma unroll
        for (int g = 0; g < 3; ++g) A[kx][g] = *reinterpret_cast<const f32x4h *>(wf + (((blk * K + kx) * 3 + g) * 64 + lane) * 4);
        A12[kx] = wf12[(blk * K + kx) * 64 + lane];
    }
    const size_t img = (size_t)H * W;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x + (size_t)b * img * x_stride), 0,
                                                                          (unsigned)(((img - 1) * x_stride + 52) * sizeof(float)), 0x00020000);
    const int row = ty * 16 + n16;
    const bool rin = row < H;
    // lane holds output channels 16 blk + 4 kq .. + 3 of row `row` for every column of the tile; accumulators start from the bias and
    // column ci - 6 is stored, straight from its accumulator, as soon as input column ci is in (see conv_col7_kernel)
    const int ch = 16 * blk + 4 * kq;
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)b * img * out_stride, 0,
                                                                           (unsigned)(((img - 1) * out_stride + 52) * sizeof(float)), 0x00020000);
    const bool chin = rin && ch < 52;
    f32x4h acc[TW];           // acc[i] is live from input column i to input column i + 6
    f32x4h b4 = *reinterpret_cast<const f32x4h *>(bias + ch);
    if (blk == 3 && kq != 0) b4 = f32x4h{0.f, 0.f, 0.f, 0.f};
    f32x4h Bq[NC][3];
    float B12[NC];
    // out-of-range columns and rows read zeros (offsets past the descriptor); channel 48 is real in lane group kq = 0 only
#define ROW_LOAD(CI)                                                                              \
    do {                                                                                          \
        const int c_ = x0 - P + (CI);                                                             \
        const bool in_ = rin & (c_ >= 0) & (c_ < W);                                              \
        const unsigned pix_ = (unsigned)((row * W + c_) * x_stride) * 4u;                         \
        const unsigned off_ = in_ ? pix_ + 16u * kq : 0x80000000u;                                \
        const unsigned o12_ = (in_ & (kq == 0)) ? pix_ + 192u : 0x80000000u;                      \
        Bq[CI][0] = __builtin_bit_cast(f32x4h, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off_, 0, 0)); \
        Bq[CI][1] = __builtin_bit_cast(f32x4h, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off_, 64, 0)); \
        Bq[CI][2] = __builtin_bit_cast(f32x4h, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off_, 128, 0)); \
        B12[CI] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)o12_, 0, 0)); \
    } while (0)
#pragma unroll
    for (int ci = 0; ci < AHEAD; ++ci) ROW_LOAD(ci);
#ifdef PIVLFN_STAMPS
    if (stamps) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); R7_NOW(ts1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); R7_NOW(ts2); }
#endif
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {
        if (ci + AHEAD < NC) ROW_LOAD(ci + AHEAD);
        if (ci < TW) acc[ci] = b4;
        __builtin_amdgcn_sched_barrier(0);
        if (blk == 3) {
            // channel 48 on the vector unit: the fragment registers hold its weights (replicated over the slots at pack time); every
            // lane multiplies its 13 channels of the column, the four channel groups of a pixel are added across lanes at the end
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int xo = ci - kx;
                if (xo < 0 || xo >= TW) continue;
                float e = acc[xo][0];
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int j = 0; j < 4; ++j) e = fmaf(A[kx][g][j], Bq[ci][g][j], e);
                e = fmaf(A12[kx], B12[ci], e);
                acc[xo][0] = e;
            }
        } else {
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) {       // consecutive MFMAs target different accumulators
                        const int xo = ci - kx;
                        if (xo < 0 || xo >= TW) continue;
                        acc[xo] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[kx][g][j], Bq[ci][g][j], acc[xo], 0, 0, 0);
                    }
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int xo = ci - kx;
                if (xo < 0 || xo >= TW) continue;
                acc[xo] = __builtin_amdgcn_mfma_f32_16x16x4f32(A12[kx], B12[ci], acc[xo], 0, 0, 0);
            }
        }
        if (ci >= K - 1) {                     // output column ci - 6 is complete
            const int xo = ci - (K - 1);
            if (blk == 3) {
                float e = acc[xo][0];
                e = e + __shfl_xor(e, 16);
                e = e + __shfl_xor(e, 32);
                acc[xo] = f32x4h{e, 0.f, 0.f, 0.f};
            }
            const int ox = x0 + xo;
            const unsigned off = (chin && ox < W) ? (unsigned)((row * W + ox) * out_stride + ch) * 4u : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4h, acc[xo]), rso, (int)off, 0, 0);
        }
    }
#undef ROW_LOAD
#ifdef PIVLFN_STAMPS
    if (stamps) R7_NOW(ts3);
#endif
#ifdef PIVLFN_STAMPS
    if (stamps && lane == 0) {
        unsigned long long *o = stamps + ((size_t)(blockIdx.x * 4 + wave) * 4 + ph) * 8;
        unsigned long long ts4;
        R7_NOW(ts4);
        o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = ts3; o[4] = ts4; o[5] = blk;
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        o[6] = hw;
    }
#endif
    }       // tiles of the workgroup
}

// wf: [4 blocks][7][3][64][4] fragments of channel quads kq + 4 g, wf12: [4][7][64] = channel 48 in lane group 0 (net.hip pack_conv);
// block 3 of both holds output channel 48's weights replicated over the slots; bias: [64]
int launch_conv_row7(const float *x, int x_stride, const float *wf, const float *wf12, const float *bias, float *out, int out_stride,
                     int B, int H, int W, hipStream_t st)
{
    PIV_REQUIRE(x && wf && wf12 && bias && out && B > 0 && H > 0 && W > 0, "conv_row7: bad arguments");
    PIV_REQUIRE(x_stride % 4 == 0 && x_stride >= 52 && out_stride % 4 == 0 && out_stride >= 52, "conv_row7: 52 stored lanes in and out (strides %d, %d)", x_stride, out_stride);
    PIV_REQUIRE((long)H * W * std::max(x_stride, out_stride) * 4 < (1L << 31), "conv_row7: image exceeds 2 GiB (32-bit buffer offsets for the loads and the stores)");
    constexpr int TW = 16;
    hipLaunchKernelGGL((conv_row7_kernel<TW>), dim3(cdiv(cdiv(W, TW) * cdiv(H, 16) * B, 4)), dim3(256), 0, st, x, x_stride, wf, wf12, bias, out, out_stride, B, H, W,
                       reinterpret_cast<unsigned long long *>(((unsigned long long)(unsigned)PIV_KNOB(6) << 32) | (unsigned)PIV_KNOB(5)));   // tools build only
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

template <int K>
static int launch_head_t(const float *x, const float *w, float b0, float b1, const float *res4, float *out4, int B, int H, int W,
                         hipStream_t st)
{
    constexpr int PW = 16 + K - 1;
    const size_t lds = ((size_t)PW * PW * 32 + K * K * 64) * sizeof(float);
    static LdsAttr attr_a, attr_b;
    if (lds > 64 * 1024) {
        if (int rc = ensure_dyn_lds(attr_a, reinterpret_cast<const void *>(conv_head_kernel<K, false>), (int)lds)) return rc;
        if (int rc = ensure_dyn_lds(attr_b, reinterpret_cast<const void *>(conv_head_kernel<K, true>), (int)lds)) return rc;
    }
    const int nblk = cdiv(W, 16) * cdiv(H, 16) * B;
    // Kernel choice per IMAGE size, never per batch: a pair's flow must not depend on its batch mates (the four-row kernel sums
    // in a different order than the two one-pixel variants, which produce the same bits as each other).
    // 7 x 7 heads on images of at least 256 x 256: the matrix-core formulation (per image, never per batch)
    if (K == 7 && (long)H * W >= 256 * 256 && (long)H * W * 32 * 4 < (1L << 31) && !(PIV_KNOB(1) & 65536 * 8)) {
        constexpr int TWO = 64 - 2 * (K / 2);
        constexpr int THM = 8;
        const size_t ldsm = (size_t)(THM / 2) * 64 * 16 * sizeof(float);
        static LdsAttr attr_m;
        if (int rc = ensure_dyn_lds(attr_m, reinterpret_cast<const void *>(conv_head_mfma_kernel<K, THM>), (int)ldsm)) return rc;
        hipLaunchKernelGGL((conv_head_mfma_kernel<K, THM>), dim3(cdiv(W, TWO) * cdiv(H, THM) * B), dim3(256), ldsm, st, x, w, b0, b1, res4, out4, B, H, W);
        PIV_CHECK_HIP(hipGetLastError());
        return PIVLFN_OK;
    }
    if ((long)H * W >= 512 * 512 && !(PIV_KNOB(1) & 8192)) {
        constexpr int PW4 = 32 + K - 1;
        hipLaunchKernelGGL((conv_head4_kernel<K>), dim3(cdiv(W, 32) * cdiv(H, 32) * B), dim3(256), (size_t)PW4 * PW4 * 8 * sizeof(float), st,
                           x, w, b0, b1, res4, out4, B, H, W);
        PIV_CHECK_HIP(hipGetLastError());
        return PIVLFN_OK;
    }
    if (cdiv(W, 16) * cdiv(H, 16) <= 512 && !(PIV_KNOB(1) & 1024))
        hipLaunchKernelGGL((conv_head_kernel<K, true>), dim3(nblk), dim3(256), lds, st, x, w, b0, b1, res4, out4, B, H, W);
    else
        hipLaunchKernelGGL((conv_head_kernel<K, false>), dim3(nblk), dim3(256), lds, st, x, w, b0, b1, res4, out4, B, H, W);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

// w: device, [k*k][8][4][2] = (tap, channel quad, channel-in-quad, output), then [k][2][64][4] = the matrix-core head's A fragments
int launch_conv_head(const float *x, const float *w, float b0, float b1, const float *res4, float *out4, int B, int H, int W,
                     int k, hipStream_t st)
{
    PIV_REQUIRE(x && w && out4 && B > 0 && H > 0 && W > 0, "conv_head: bad arguments");
    switch (k) {
        case 3: return launch_head_t<3>(x, w, b0, b1, res4, out4, B, H, W, st);
        case 5: return launch_head_t<5>(x, w, b0, b1, res4, out4, B, H, W, st);
        case 7: return launch_head_t<7>(x, w, b0, b1, res4, out4, B, H, W, st);
    }
    set_error("conv_head: k=%d unsupported", k);
    return PIVLFN_ERR_ARG;
}

}  // namespace pivlfn
