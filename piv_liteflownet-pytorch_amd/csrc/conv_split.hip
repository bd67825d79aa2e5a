// fp32 convolution on the fp16 matrix cores by exact operand splitting.
//
// gfx950's fp32 matrix instruction (v_mfma_f32_32x32x2_f32, conv_mfma.hip) runs at 1/16 of the rate of
// v_mfma_f32_32x32x16_f16.  An fp32 number is the exact sum of three fp16 numbers at staggered scales,
//
//     x = h + m * 2^-11 + l * 2^-22,     h = rn16(x),  m = rn16((x - h) * 2^11),  l = rn16(((x - h) * 2^11 - m) * 2^11)
//
// (11 + 11 + 2 significand bits; every subtraction is exact in fp32; exact for 2^-14 <= |x| < 65504 -- below 2^-14 the conversion
// flushes h and the other pieces carry 22 bits: |error| < 2^-37), so a product of two fp32 numbers is the sum of nine
// fp16 x fp16 products, each exact in the fp32 the matrix core accumulates in.  Six of them carry everything above 2^-33 of the
// product:  h.h | h.m 2^-11, m.h 2^-11 | m.m 2^-22, h.l 2^-22, l.h 2^-22;  the three dropped ones are <= 2^-32 relative -- 256 x
// below the rounding an fp32 fma chain itself commits per step.  The sum over K and over the six terms is accumulated in fp32
// by the instruction, exactly as the fp32 instruction does for its products.  6 fp16 MFMAs replace 8 fp32 MFMAs of 1/16 the
// rate: 2.67 x the matrix throughput at fp32 accuracy (tests/test_gpu_split.py: the error against a float64 convolution is that
// of conv_mfma.hip).  Inputs, outputs, bias and activation are fp32; |x| must stay below 65504 (fp16 range of the head piece).
//
// Scales.  The 2^-11 / 2^-22 factors have to live in an operand (one accumulator).  Activation magnitudes are not known, so
// the activation pieces stay at their natural scale (h, m, l all of the magnitude of x: no fp16 underflow) and the factors go
// to the weight side, whose magnitudes are known at load time: a layer's weights are scaled by a power of two so that max |w|
// lands in [2^13, 2^14) and the result is scaled back (exactly) in the epilogue.  Per weight the kernel multiplies
//     h by  P0 = bh,  P1 = bm 2^-11,  P2 = bl 2^-22        (three stored pieces)
//     m by  P0 2^-11, P1 2^-11                              (derived in registers: v_pk_mul_f16 by a power of two)
//     l by  P0 2^-22
// Pieces that fall below 2^-14 become fp16 subnormals (or zero); what they then lose is below 2^-25 of the *largest* weight
// times the activation -- 2^-35 of a typical sum.
//
// Shape: conv_f16.hip's (workgroup = 4 waves = (4*MT rows x 32 px) x (32*NT channels), K chunk = 16 channels = one MFMA
// k-step per tap and piece pair, patch staged once per chunk, next chunk prefetched global -> registers under the MFMAs),
// except that the weight slab of a chunk (3 pieces) is staged one kernel row at a time, so that two workgroups fit a CU
// and alternate on the matrix pipe while the other splits and stages:
//   LDS pixel record = 3 pieces x 16 halfs + 8 halfs of padding = 112 bytes (7 * pixel mod 16 visits all sixteen 16-byte
//   slots of a 256-byte bank row: conflict-free ds_read_b128); weights [kx][piece][k-half][channel][8].
#include <cmath>
#include <cstring>
#include <vector>
#include "common.h"

namespace pivlfn {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));


// x -> (h, m, l) with x = h + m 2^-11 + l 2^-22 exactly (|x| < 65504, pieces not subnormal)
__device__ __forceinline__ void split3(float x, _Float16 &h, _Float16 &m, _Float16 &l)
{
    h = (_Float16)x;
    const float r1 = (x - (float)h) * 2048.f;
    m = (_Float16)r1;
    const float r2 = (r1 - (float)m) * 2048.f;
    l = (_Float16)r2;
}

// TERMS = 6: all six partial products (three pieces per operand).  TERMS = 3: h.h, h.m, m.h only (two pieces per operand, 80-byte
// pixel records): each product then carries a relative error of up to 2^-21 (typically 2^-23.5) instead of 2^-32.
template <int TERMS, int MT, int NT, int PM, int WM, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_split_kernel(const ConvParamsX p)
{
    extern __shared__ __attribute__((aligned(16))) _Float16 xsmem[];
    constexpr int NP = TERMS == 6 ? 3 : 2;       // pieces per operand
    constexpr int XPITCH = NP * 16 + 8;          // halfs per staged pixel (112 / 80 bytes: 7 or 5 sixteen-byte slots, odd -> conflict-free)
    constexpr int BN = NT * 32;
    constexpr int TH = 4 * MT;
    const int PH = (TH - 1) * p.S + p.KH;
    const int PW = 31 * p.S + p.KW;
    const int npix = PH * PW;
    _Float16 *patch = xsmem;
    _Float16 *wts = xsmem + npix * XPITCH;                                   // [KW][NP][2][BN][8]
    float *lbias = reinterpret_cast<float *>(wts + p.KW * NP * 2 * BN * 8); // this workgroup's BN biases

    const int tiles_x = (p.Wo + 31) >> 5;
    const int tiles_y = (p.Ho + TH - 1) / TH;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int n0 = blockIdx.y * BN;
    const int x0 = tx * 32, y0 = ty * TH;
    const int ix0 = x0 * p.S - p.padX, iy0 = y0 * p.S - p.padY;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int row = lane & 31, hh = lane >> 5;
    const int sub = tid & 1;

    if (tid < BN) lbias[tid] = p.bias[n0 + tid];       // visible after the first barrier of the K loop

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    int abase[MT];      // half offset of this lane's pixel operand (piece 0) for tap (0,0)
#pragma unroll
    for (int m = 0; m < MT; ++m) abase[m] = ((wave * MT + m) * p.S * PW + row * p.S) * XPITCH + hh * 8;
    const int bbase = (hh * BN + row) * 8;

    // staging items: patch item i = pixel (tid>>1) + 128*i, channel half `sub` (channels 4 sub .. +3 and 8 + 4 sub .. +3);
    // weight item i = 16-byte unit tid + 256*i of the row's [KW][3][2][BN] units
    int poff[PM];       // pixel index inside the source, -1 = outside the image, -2 = no item
#pragma unroll
    for (int i = 0; i < PM; ++i) {
        const int pix = (tid >> 1) + 128 * i;
        const int py = pix / PW, px = pix - py * PW;
        const int iy = iy0 + py, ix = ix0 + px;
        poff[i] = pix >= npix ? -2 : ((iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? (b * p.H + iy) * p.W + ix : -1);
    }
    const int nw = p.KW * NP * 2 * BN;
    const f32x4 *wsrc = reinterpret_cast<const f32x4 *>(p.wpk) + n0;
    const size_t wrow = (size_t)p.KW * 6 * p.cout_pad;      // 16-byte units per (chunk, ky) in the packed weights (always 3 pieces)

    f32x4 pr[2 * PM], wr[WM];
    int seg = 0, c0 = 0;
    const float *sp = p.seg[0].ptr;
    int scl = p.seg[0].cload, sst = p.seg[0].stride;

#define X_LOAD_PATCH()                                                                            \
    do {                                                                                          \
        if (PIV_DBG(p) & 2) break;                                                                     \
        const bool ok0_ = c0 + 4 * sub < scl, ok1_ = c0 + 8 + 4 * sub < scl;                      \
        _Pragma("unroll") for (int i = 0; i < PM; ++i) {                                          \
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};                           \
            if (poff[i] >= 0) {                                                                   \
                const float *q_ = sp + (size_t)poff[i] * sst + c0 + 4 * sub;                      \
                if (ok0_) v0 = *reinterpret_cast<const f32x4 *>(q_);                              \
                if (ok1_) v1 = *reinterpret_cast<const f32x4 *>(q_ + 8);                          \
            }                                                                                     \
            pr[2 * i] = v0;                                                                       \
            pr[2 * i + 1] = v1;                                                                   \
        }                                                                                         \
    } while (0)
#define X_LOAD_W(ROWIDX)                                                                          \
    do {                                                                                          \
        if (PIV_DBG(p) & 2) break;                                                                     \
        const f32x4 *wc_ = wsrc + (size_t)(ROWIDX)*wrow;                                          \
        _Pragma("unroll") for (int i = 0; i < WM; ++i) {                                          \
            const int idx_ = tid + 256 * i;                                                       \
            const int r_ = idx_ / BN;       /* staged row (kx * NP + piece) * 2 + k-half */        \
            const int src_ = NP == 3 ? r_ : (r_ / 4) * 6 + (r_ % 4);                               \
            if (idx_ < nw) wr[i] = wc_[src_ * p.cout_pad + (idx_ % BN)];                          \
        }                                                                                         \
    } while (0)

    const h8 k11 = {(_Float16)0x1p-11f, (_Float16)0x1p-11f, (_Float16)0x1p-11f, (_Float16)0x1p-11f,
                    (_Float16)0x1p-11f, (_Float16)0x1p-11f, (_Float16)0x1p-11f, (_Float16)0x1p-11f};
    const h8 k22 = {(_Float16)0x1p-22f, (_Float16)0x1p-22f, (_Float16)0x1p-22f, (_Float16)0x1p-22f,
                    (_Float16)0x1p-22f, (_Float16)0x1p-22f, (_Float16)0x1p-22f, (_Float16)0x1p-22f};

    // this workgroup's K range: all chunks, or with split-K (gridDim.z > 1) an even share of them
    const int nz = gridDim.z, kz = blockIdx.z;
    const int c_begin = (int)((long)p.nchunk * kz / nz), c_end = (int)((long)p.nchunk * (kz + 1) / nz);
    for (int c = 0; c < c_begin; ++c) {     // walk the sources to the first chunk of the share
        c0 += 16;
        if (c0 >= scl) {
            ++seg;
            c0 = 0;
            sp = p.seg[seg].ptr;
            scl = p.seg[seg].cload; sst = p.seg[seg].stride;
        }
    }
    X_LOAD_PATCH();
    X_LOAD_W(c_begin * p.KH);
    const int nrows = c_end * p.KH;         // (chunk, ky) phases, absolute row index into the packed weights
    int phase = c_begin * p.KH;
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
        // registers -> LDS: split this chunk's fp32 patch into its three fp16 pieces (the previous chunk's last barrier has
        // been passed by every wave: the patch buffer is free)
        if (!(PIV_DBG(p) & 4)) {
#pragma unroll
            for (int i = 0; i < PM; ++i)
                if (poff[i] != -2) {
                    _Float16 *d = patch + ((tid >> 1) + 128 * i) * XPITCH + 4 * sub;
                    h4 ph[2], pm[2], pl[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const f32x4 v = pr[2 * i + e];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            _Float16 a, bq, c;
                            split3(v[j], a, bq, c);
                            ph[e][j] = a; pm[e][j] = bq; pl[e][j] = c;
                        }
                    }
                    *reinterpret_cast<h4 *>(d) = ph[0];
                    *reinterpret_cast<h4 *>(d + 8) = ph[1];
                    *reinterpret_cast<h4 *>(d + 16) = pm[0];
                    *reinterpret_cast<h4 *>(d + 24) = pm[1];
                    if (NP == 3) {
                        *reinterpret_cast<h4 *>(d + 32) = pl[0];
                        *reinterpret_cast<h4 *>(d + 40) = pl[1];
                    }
                }
        }
        for (int ky = 0; ky < p.KH; ++ky, ++phase) {
            if (!(PIV_DBG(p) & 4)) {
#pragma unroll
                for (int i = 0; i < WM; ++i)
                    if (tid + 256 * i < nw) reinterpret_cast<f32x4 *>(wts)[tid + 256 * i] = wr[i];
            }
            __syncthreads();
            if (ky == 0 && chunk + 1 < c_end) {
                c0 += 16;
                if (c0 >= scl) {
                    ++seg;
                    c0 = 0;
                    sp = p.seg[seg].ptr;
                    scl = p.seg[seg].cload; sst = p.seg[seg].stride;
                }
                X_LOAD_PATCH();
            }
            if (phase + 1 < nrows) X_LOAD_W(phase + 1);
            if (!(PIV_DBG(p) & 1)) {
                for (int kx = 0; kx < p.KW; ++kx) {
                    const int toff = (ky * PW + kx) * XPITCH;
                    h8 a[MT][NP];
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int j = 0; j < NP; ++j) a[m][j] = *reinterpret_cast<const h8 *>(patch + abase[m] + toff + j * 16);
                    const _Float16 *wk = wts + kx * NP * 2 * BN * 8 + bbase;
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const h8 w0 = *reinterpret_cast<const h8 *>(wk + n * 256);
                        const h8 w1 = *reinterpret_cast<const h8 *>(wk + 2 * BN * 8 + n * 256);
                        const h8 w0m = w0 * k11;
#pragma unroll
                        for (int m = 0; m < MT; ++m) {      // A = channels, B = pixels; smallest terms first
                            if (NP == 3) {
                                const h8 w2 = *reinterpret_cast<const h8 *>(wk + 4 * BN * 8 + n * 256);
                                const h8 w1m = w1 * k11, w0l = w0 * k22;
                                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0l, a[m][NP - 1], acc[m][n], 0, 0, 0);
                                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, a[m][0], acc[m][n], 0, 0, 0);
                                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1m, a[m][1], acc[m][n], 0, 0, 0);
                            }
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0m, a[m][1], acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, a[m][0], acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, a[m][0], acc[m][n], 0, 0, 0);
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
#undef X_LOAD_PATCH
#undef X_LOAD_W

    // Epilogue: lane&31 = pixel, registers 4g..4g+3 = channels 8g + 4*hh + {0..3} (the D layout of the 32x32 instructions).
    if (nz > 1) {
        // split-K: this share's partial sums (already scaled back) to scratch[z][pixel][cout_pad]; conv_splitk_reduce_kernel adds
        // the shares in z order, then bias and activation
        const int ox = x0 + row;
        const size_t npix = (size_t)p.B * p.Ho * p.Wo;
        const float sc = p.out_scale;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int oy = y0 + wave * MT + m;
            if (oy >= p.Ho || ox >= p.Wo) continue;
            float *dst = p.scratch + ((size_t)kz * npix + (size_t)(b * p.Ho + oy) * p.Wo + ox) * p.cout_pad;
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = n0 + n * 32 + 8 * g + 4 * hh;
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[m][n][4 * g + j] * sc;
                    *reinterpret_cast<f32x4 *>(dst + ch) = v;
                }
        }
        return;
    }
    {
        const int ox = x0 + row;
        const bool interior = x0 + 32 <= p.Wo && y0 + TH <= p.Ho && n0 + BN <= p.cout_store;
        const float sc = p.out_scale;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int oy = y0 + wave * MT + m;
            const bool pix_ok = interior || (oy < p.Ho && ox < p.Wo);
            const size_t pix = (size_t)(b * p.Ho + (oy < p.Ho ? oy : 0)) * p.Wo + (ox < p.Wo ? ox : 0);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = n0 + n * 32 + 8 * g + 4 * hh;
                    if (!pix_ok || (!interior && ch >= p.cout_store)) continue;
                    const f32x4 bq = *reinterpret_cast<const f32x4 *>(lbias + (ch - n0));
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[m][n][4 * g + j] * sc + bq[j];     // sc is a power of two: exact
                    if (p.lrelu) {
                        v[0] = lrelu01(v[0]); v[1] = lrelu01(v[1]); v[2] = lrelu01(v[2]); v[3] = lrelu01(v[3]);
                    }
                    *reinterpret_cast<f32x4 *>(p.out + pix * p.out_stride + ch) = v;
                }
            }
        }
    }
}

template <int TERMS, int MT, int NT, int PM, int WM, int OCC>
static int launch_x(const ConvParamsX &p, hipStream_t st)
{
    constexpr int TH = 4 * MT, BN = NT * 32, NP = TERMS == 6 ? 3 : 2, XPITCH = NP * 16 + 8;
    const int PH = (TH - 1) * p.S + p.KH, PW = 31 * p.S + p.KW;
    const size_t lds = ((size_t)PH * PW * XPITCH + (size_t)p.KW * NP * 2 * BN * 8) * sizeof(_Float16) + BN * sizeof(float);
    PIV_REQUIRE(lds * OCC <= 160 * 1024, "conv_split: LDS tile of %zu bytes x %d exceeds 160 KiB (k=%dx%d s=%d)", lds, OCC, p.KH, p.KW, p.S);
    PIV_REQUIRE(PH * PW * 2 <= 256 * PM && p.KW * NP * 2 * BN <= 256 * WM, "conv_split: internal staging bound exceeded");
    static LdsAttr attr;
    if (int rc = ensure_dyn_lds(attr, reinterpret_cast<const void *>(conv_split_kernel<TERMS, MT, NT, PM, WM, OCC>), 160 * 1024)) return rc;
    const int tiles = cdiv(p.Wo, 32) * cdiv(p.Ho, TH) * p.B;
    const int nz = p.ksplit > 1 ? p.ksplit : 1;
    dim3 grid(tiles, p.cout_pad / BN, nz);
    hipLaunchKernelGGL((conv_split_kernel<TERMS, MT, NT, PM, WM, OCC>), grid, dim3(256), lds, st, p);
    PIV_CHECK_HIP(hipGetLastError());
    if (nz > 1) {       // second pass: the fp32 kernel's reduction (conv_mfma.hip)
        ConvParams r;
        memset(&r, 0, sizeof(r));
        r.scratch = p.scratch; r.bias = p.bias; r.out = p.out; r.out_stride = p.out_stride; r.cout_store = p.cout_store;
        r.cout_pad = p.cout_pad; r.B = p.B; r.Ho = p.Ho; r.Wo = p.Wo; r.lrelu = p.lrelu;
        return launch_splitk_reduce(r, nz, st);
    }
    return PIVLFN_OK;
}

// ---- 16-row tile, one workgroup per CU, everything double-buffered (three-term products, 3 x 3 stride 1) ---------------------
// conv_split_kernel's two workgroups per CU each alternate staging -> barrier -> MFMAs -> barrier: the matrix pipe idles whenever
// both are outside their MFMA block.  Here one workgroup of four waves owns the CU and a wave never leaves its MFMA stream:
//   * a wave computes 4 rows x 32 px x 32 NT channels: 16 LDS operand reads per 12 NT MFMAs and tap (NT = 4: 0.33 per MFMA
//     instead of 0.5), 64 NT accumulator registers out of the 512 a wave has at this occupancy;
//   * patch and weight rows are double-buffered in LDS: what a phase (chunk c, kernel row ky) stages -- its share of chunk c+1's
//     patch, split into fp16 pieces, and weight row ph+1 -- is only read after the next barrier; one barrier per phase;
//   * the operand fragments of tap kx+1 are read from LDS before the MFMAs of tap kx are issued (two register sets);
//   * the phase body is straight-line code (no branches: out-of-image pixels and channel tails are buffer loads with an
//     out-of-range offset, which return zero; past-the-end stages are clamped to the last chunk / row and land in a buffer nobody
//     reads), so that the compiler can place the staging VALU / LDS / memory instructions between the MFMAs;
//   * a trailing 4-lane source (the 49-, 130-, 131-channel inputs) is not a 16-channel chunk that is 3/4 zeros but one extra phase
//     with the taps folded into K (three K16 steps of 4 taps x 4 channels; p.tail / p.wtail).
constexpr unsigned XOOB = 0x80000000u;

template <int NT>
__global__ __launch_bounds__(256, 1) void conv_split_big_kernel(const ConvParamsX p)
{
    extern __shared__ __attribute__((aligned(16))) _Float16 xsmem[];
    constexpr int XPITCH = 40;                   // two pieces x 16 channels + 8 halfs of padding (80 bytes)
    constexpr int BN = NT * 32;
    constexpr int MT = 4, TH = 16, PH = 18, PW = 34, NPIX = PH * PW;       // 612 staged pixels
    constexpr int PM = 5;                        // patch items per thread: pixel (tid >> 1) + 128 i, channel half tid & 1
    constexpr int WROW = 3 * 2 * 2 * BN * 8;     // halfs per staged weight row [kx][piece][k-half][BN][8]
    constexpr int WM = 3 * 2 * 2 * BN / 256;     // 16-byte weight units per thread and row (6 / 3)
    constexpr int PBUF = (NPIX + 1) * XPITCH;    // one patch buffer: the staged pixels + a spare record (target of items that do not exist)
    _Float16 *patch0 = xsmem;
    _Float16 *wts0 = xsmem + 2 * PBUF;
    float *lbias = reinterpret_cast<float *>(wts0 + 2 * WROW);

    const int tiles_x = (p.Wo + 31) >> 5;
    const int tiles_y = (p.Ho + TH - 1) / TH;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int n0 = blockIdx.y * BN;
    const int x0 = tx * 32, y0 = ty * TH;
    const int ix0 = x0 - p.padX, iy0 = y0 - p.padY;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = lane & 31, hh = lane >> 5;
    const int sub = tid & 1;

    if (tid < BN) lbias[tid] = p.bias[n0 + tid];       // visible after the first barrier

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    int abase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) abase[m] = ((wave * MT + m) * PW + row) * XPITCH + hh * 8;
    const int bbase = (hh * BN + row) * 8;

    // staged pixel -> byte offset inside its image's source record (times the pixel stride later), XOOB = outside / no item
    unsigned ppix[PM];
    int pdst[PM];            // LDS half offset of the item's record (the spare record for items that do not exist)
#pragma unroll
    for (int i = 0; i < PM; ++i) {
        const int pix = (tid >> 1) + 128 * i;
        const int py = pix / PW, px = pix - py * PW;
        const int iy = iy0 + py, ix = ix0 + px;
        const bool in = pix < NPIX && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        ppix[i] = in ? (unsigned)(iy * p.W + ix) : XOOB;
        pdst[i] = (pix < NPIX ? pix : NPIX) * XPITCH + 4 * sub;
    }
    const f32x4 *wsrc = reinterpret_cast<const f32x4 *>(p.wpk) + n0;
    const size_t wrow = (size_t)3 * 6 * p.cout_pad;      // 16-byte units per (chunk, ky) in the packed weights (3 pieces)
    int wsrc_off[WM];       // unit offset of this thread's weight items inside a packed row (pieces 0 and 1 of the 3 stored)
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx / BN;                           // staged row (kx * 2 + piece) * 2 + k-half
        wsrc_off[i] = ((r >> 2) * 6 + (r & 3)) * p.cout_pad + (idx % BN);
    }

    f32x4 pr[2 * PM], wr[WM];
    // the chunk whose patch is loaded next
    int seg = 0, c0 = 0, lchunk = 0;
    int scl = p.seg[0].cload, sst4 = p.seg[0].stride * 4;
    const size_t img_px = (size_t)p.H * p.W;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.seg[0].ptr + (size_t)b * img_px * p.seg[0].stride), 0,
                                                                    (unsigned)(((img_px - 1) * p.seg[0].stride + p.seg[0].cload) * 4), 0x00020000);
#define XB_ADVANCE()                                                                              \
    do {                                                                                          \
        if (lchunk + 1 < p.nchunk) {                                                              \
            ++lchunk;                                                                             \
            c0 += 16;                                                                             \
            if (c0 >= scl) {                                                                      \
                ++seg;                                                                            \
                c0 = 0;                                                                           \
                scl = p.seg[seg].cload; sst4 = p.seg[seg].stride * 4;                             \
                rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.seg[seg].ptr + (size_t)b * img_px * p.seg[seg].stride), 0, \
                                                       (unsigned)(((img_px - 1) * p.seg[seg].stride + p.seg[seg].cload) * 4), 0x00020000); \
            }                                                                                     \
        }                                                                                         \
    } while (0)
#define XB_LOAD_ITEM(I)                                                                           \
    do {                                                                                          \
        const unsigned base_ = ppix[I] == XOOB ? XOOB : ppix[I] * (unsigned)sst4 + (unsigned)(c0 + 4 * sub) * 4u; \
        const unsigned o0_ = (c0 + 4 * sub < scl) ? base_ : XOOB;                                 \
        const unsigned o1_ = (c0 + 8 + 4 * sub < scl && base_ != XOOB) ? base_ + 32u : XOOB;      \
        pr[2 * (I)] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)o0_, 0, 0));      \
        pr[2 * (I) + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)o1_, 0, 0));  \
    } while (0)
// TL (uniform): the chunk being staged is the folded 4-channel tail -- 16-byte pixel records (h4 | m4) written by the lanes that
// hold channels 0..3 (sub == 0); everything else goes to the spare record.  Selects, not branches: the phases stay straight-line.
#define XB_COMMIT_ITEM(I, DST, TL)                                                                \
    do {                                                                                          \
        const int pixd_ = (pdst[I] - 4 * sub) / XPITCH;      /* = the item's pixel, NPIX for items that do not exist */ \
        const int tdst_ = (sub == 0 && pixd_ < NPIX) ? pixd_ * 8 : NPIX * XPITCH;                 \
        _Float16 *d_ = (DST) + ((TL) ? NPIX * XPITCH : pdst[I]);                                   \
        _Float16 *dt_ = (DST) + ((TL) ? tdst_ : NPIX * XPITCH + 32);                               \
        h4 ph_[2], pm_[2];                                                                        \
        _Pragma("unroll") for (int e_ = 0; e_ < 2; ++e_) {                                        \
            const f32x4 v_ = pr[2 * (I) + e_];                                                    \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                    \
                const _Float16 a_ = (_Float16)v_[j_];                                             \
                ph_[e_][j_] = a_;                                                                 \
                pm_[e_][j_] = (_Float16)((v_[j_] - (float)a_) * 2048.f);                          \
            }                                                                                     \
        }                                                                                         \
        *reinterpret_cast<h4 *>(d_) = ph_[0];                                                     \
        *reinterpret_cast<h4 *>(d_ + 8) = ph_[1];                                                 \
        *reinterpret_cast<h4 *>(d_ + 16) = pm_[0];                                                \
        *reinterpret_cast<h4 *>(d_ + 24) = pm_[1];                                                \
        *reinterpret_cast<h4 *>(dt_) = ph_[0];                                                    \
        *reinterpret_cast<h4 *>(dt_ + 4) = pm_[0];                                                \
    } while (0)
// rows 0 .. 3 nfull - 1 are kernel rows of full chunks; row 3 nfull (p.tail) is the tail's slab [step][piece][k-half][cout_pad] --
// two pieces per step packed densely, so a unit's offset is the normal one minus 2 (r >> 2) cout_pad, r >> 2 = its step
#define XB_LOAD_W(ROWIDX)                                                                         \
    do {                                                                                          \
        const bool tl_ = p.tail && (ROWIDX) >= 3 * nfull;                                         \
        const f32x4 *wc_ = tl_ ? wtail : wsrc + (size_t)(ROWIDX)*wrow;                            \
        const int back_ = tl_ ? 2 * p.cout_pad : 0;                                               \
        _Pragma("unroll") for (int i = 0; i < WM; ++i) wr[i] = wc_[wsrc_off[i] - ((256 * i) / (4 * BN)) * back_ - (tid / (4 * BN)) * back_]; \
    } while (0)
#define XB_COMMIT_W(DST)                                                                          \
    do {                                                                                          \
        _Pragma("unroll") for (int i = 0; i < WM; ++i) reinterpret_cast<f32x4 *>(DST)[tid + 256 * i] = wr[i]; \
    } while (0)
#define XB_FRAGS(KY, KX, FA, FW)                                                                  \
    do {                                                                                          \
        const int toff_ = ((KY)*PW + (KX)) * XPITCH;                                              \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                          \
            FA[m][0] = *reinterpret_cast<const h8 *>(patch + abase[m] + toff_);                   \
            FA[m][1] = *reinterpret_cast<const h8 *>(patch + abase[m] + toff_ + 16);              \
        }                                                                                         \
        const _Float16 *wk_ = wts + (KX)*4 * BN * 8 + bbase;                                      \
        _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                          \
            FW[n][0] = *reinterpret_cast<const h8 *>(wk_ + n * 256);                              \
            FW[n][1] = *reinterpret_cast<const h8 *>(wk_ + 2 * BN * 8 + n * 256);                 \
        }                                                                                         \
    } while (0)
#define XB_TAP(FA, FW)                                                                            \
    do {                                                                                          \
        _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                          \
            const h8 w0m_ = FW[n][0] * k11;                                                       \
            _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                      \
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0m_, FA[m][1], acc[m][n], 0, 0, 0);     \
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FW[n][1], FA[m][0], acc[m][n], 0, 0, 0); \
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FW[n][0], FA[m][0], acc[m][n], 0, 0, 0); \
            }                                                                                     \
        }                                                                                         \
    } while (0)
// phase KY of the current chunk: items I0 (and I1, -1 = none) of the next chunk's patch are staged in it
#define XB_PHASE(KY, I0, I1)                                                                      \
    do {                                                                                          \
        const _Float16 *wts = wts0 + (ph & 1) * WROW;                                             \
        _Float16 *wts_next = wts0 + ((ph + 1) & 1) * WROW;                                        \
        h8 fa0[MT][2], fw0[NT][2], fa1[MT][2], fw1[NT][2];                                        \
        XB_FRAGS(KY, 0, fa0, fw0);                                                                \
        XB_FRAGS(KY, 1, fa1, fw1);                                                                \
        XB_TAP(fa0, fw0);                                                                         \
        XB_COMMIT_W(wts_next);                                                                    \
        XB_COMMIT_ITEM(I0, patch_next, tl_next);                                                  \
        if ((I1) >= 0) XB_COMMIT_ITEM((I1) >= 0 ? (I1) : 0, patch_next, tl_next);                 \
        XB_FRAGS(KY, 2, fa0, fw0);                                                                \
        XB_TAP(fa1, fw1);                                                                         \
        XB_LOAD_W(ph + 2 < nph ? ph + 2 : nph - 1);                                               \
        XB_LOAD_ITEM(I0);                                                                         \
        if ((I1) >= 0) XB_LOAD_ITEM((I1) >= 0 ? (I1) : 0);                                        \
        XB_TAP(fa0, fw0);                                                                         \
        __syncthreads();                                                                          \
        ++ph;                                                                                     \
    } while (0)

    const h8 k11 = {(_Float16)0x1p-11f, (_Float16)0x1p-11f, (_Float16)0x1p-11f, (_Float16)0x1p-11f,
                    (_Float16)0x1p-11f, (_Float16)0x1p-11f, (_Float16)0x1p-11f, (_Float16)0x1p-11f};
    const int nfull = p.nchunk - p.tail;                 // full 16-channel chunks; a folded 4-channel tail may follow
    const int nph = nfull * 3 + p.tail;                  // weight slabs: three kernel rows per full chunk (+ the tail's)
    const f32x4 *wtail = reinterpret_cast<const f32x4 *>(p.wtail) + n0;
    // prologue: chunk 0 and weight row 0 into LDS, chunk 1 and row 1 into registers
#pragma unroll
    for (int i = 0; i < PM; ++i) XB_LOAD_ITEM(i);
    XB_LOAD_W(0);
#pragma unroll
    for (int i = 0; i < PM; ++i) XB_COMMIT_ITEM(i, patch0, false);
    XB_COMMIT_W(wts0);
    XB_ADVANCE();
#pragma unroll
    for (int i = 0; i < PM; ++i) XB_LOAD_ITEM(i);
    XB_LOAD_W(nph > 1 ? 1 : 0);
    __syncthreads();

    int ph = 0;
    for (int chunk = 0; chunk < nfull; ++chunk) {
        const _Float16 *patch = patch0 + (chunk & 1) * PBUF;
        _Float16 *patch_next = patch0 + ((chunk + 1) & 1) * PBUF;
        const bool tl_next = p.tail && chunk + 1 == nfull;       // what this chunk's phases stage is the tail
        XB_ADVANCE();            // the registers this chunk's phases free are refilled with chunk + 2 (clamped to the last chunk)
        XB_PHASE(0, 0, 1);
        XB_PHASE(1, 2, 3);
        XB_PHASE(2, 4, -1);
    }
    if (p.tail) {
        // The last source's 4 channels, taps folded into K: K16 step g covers taps 4g .. 4g+3 x 4 channels (k-half hh: taps 4g+2hh,
        // 4g+2hh+1), three steps instead of nine taps of a chunk that would be 3/4 zeros.  Its records and its weight slab were
        // staged by the last full chunk's phases.
        const _Float16 *tp = patch0 + (nfull & 1) * PBUF;
        const _Float16 *wts = wts0 + (ph & 1) * WROW;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int ta = 4 * g + 2 * hh, tb = ta + 1 < 9 ? ta + 1 : 8, tc = ta < 9 ? ta : 8;      // taps past the ninth have zero weights
            const int oa = ((tc / 3) * PW + tc % 3) * 8, ob = ((tb / 3) * PW + tb % 3) * 8;
            h8 fa[MT][2], fw[NT][2];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const _Float16 *q = tp + ((wave * MT + m) * PW + row) * 8;
                const h4 ha = *reinterpret_cast<const h4 *>(q + oa), ma = *reinterpret_cast<const h4 *>(q + oa + 4);
                const h4 hb = *reinterpret_cast<const h4 *>(q + ob), mb = *reinterpret_cast<const h4 *>(q + ob + 4);
                fa[m][0] = __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7);
                fa[m][1] = __builtin_shufflevector(ma, mb, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            const _Float16 *wk = wts + g * 4 * BN * 8 + bbase;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                fw[n][0] = *reinterpret_cast<const h8 *>(wk + n * 256);
                fw[n][1] = *reinterpret_cast<const h8 *>(wk + 2 * BN * 8 + n * 256);
            }
            XB_TAP(fa, fw);
        }
    }
#undef XB_ADVANCE
#undef XB_LOAD_ITEM
#undef XB_COMMIT_ITEM
#undef XB_LOAD_W
#undef XB_COMMIT_W
#undef XB_FRAGS
#undef XB_TAP
#undef XB_PHASE

    {
        const int ox = x0 + row;
        const bool interior = x0 + 32 <= p.Wo && y0 + TH <= p.Ho && n0 + BN <= p.cout_store;
        const float sc = p.out_scale;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int oy = y0 + wave * MT + m;
            const bool pix_ok = interior || (oy < p.Ho && ox < p.Wo);
            const size_t pix = (size_t)(b * p.Ho + (oy < p.Ho ? oy : 0)) * p.Wo + (ox < p.Wo ? ox : 0);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = n0 + n * 32 + 8 * g + 4 * hh;
                    if (!pix_ok || (!interior && ch >= p.cout_store)) continue;
                    const f32x4 bq = *reinterpret_cast<const f32x4 *>(lbias + (ch - n0));
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[m][n][4 * g + j] * sc + bq[j];
                    if (p.lrelu) {
                        v[0] = lrelu01(v[0]); v[1] = lrelu01(v[1]); v[2] = lrelu01(v[2]); v[3] = lrelu01(v[3]);
                    }
                    *reinterpret_cast<f32x4 *>(p.out + pix * p.out_stride + ch) = v;
                }
            }
        }
    }
}

template <int NT>
static int launch_xbig(const ConvParamsX &p, hipStream_t st)
{
    constexpr int BN = NT * 32;
    const size_t lds = ((size_t)2 * 613 * 40 + (size_t)2 * 3 * 4 * BN * 8) * sizeof(_Float16) + BN * sizeof(float);
    static LdsAttr attr;
    if (int rc = ensure_dyn_lds(attr, reinterpret_cast<const void *>(conv_split_big_kernel<NT>), 160 * 1024)) return rc;
    const int tiles = cdiv(p.Wo, 32) * cdiv(p.Ho, 16) * p.B;
    dim3 grid(tiles, p.cout_pad / BN);
    hipLaunchKernelGGL((conv_split_big_kernel<NT>), grid, dim3(256), lds, st, p);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

// Which layers the kernel takes (net.hip asks before routing): stride 1 with at most 7 taps per kernel row, or 3 x 3 at stride 2
// with three-term products (4-row tiles: the stride-2 patch of an 8-row tile does not fit twice per CU); 16-byte granular sources.
bool conv_split_supports(int KH, int KW, int S, int cout_pad, int terms)
{
    if (cout_pad % 32 != 0) return false;
    if (S == 2) return terms == 3 && KH == 3 && KW == 3;
    if (S != 1 || KW > 7 || KH > 7) return false;
    const int PH = 7 + KH, PW = 31 + KW;                 // 8-row tile
    const int bn = cout_pad % 64 == 0 ? 64 : 32;
    const size_t lds = ((size_t)PH * PW * 56 + (size_t)KW * 6 * bn * 8) * 2 + bn * 4;      // the six-term layout, the larger one
    return lds * 2 <= 160 * 1024 && PH * PW * 2 <= 256 * 5 && KW * 6 * bn <= 256 * 11;
}

int launch_conv_x(const ConvParamsX &p_in, hipStream_t st)
{
    ConvParamsX p = p_in;
    PIV_SET_DBG(p, PIV_KNOB(3));
    PIV_REQUIRE(p.nseg >= 1 && p.nseg <= 3 && p.nchunk >= 1, "conv_split: bad segment description");
    PIV_REQUIRE(p.cout_pad % 32 == 0 && p.cout_store <= p.cout_pad && p.cout_store % 4 == 0, "conv_split: bad output channel counts");
    for (int i = 0; i < p.nseg; ++i)
        PIV_REQUIRE(p.seg[i].ptr && p.seg[i].cload % 4 == 0 && p.seg[i].stride % 4 == 0, "conv_split: source %d must be 16-byte granular", i);
    PIV_REQUIRE(p.Ho > 0 && p.Wo > 0 && p.B > 0, "conv_split: empty output");
    // tools-build knobs (pivlfn_tune(1, .); all zero in libpivlfn.so): 65536 force three terms, 131072 / 262144 no 16-row kernel for
    // 64 / 128 channels, 524288 no 16-row small-kernel tiles for 64 channels
    if (PIV_KNOB(1) & 65536) p.terms = 3;
    PIV_REQUIRE(conv_split_supports(p.KH, p.KW, p.S, p.cout_pad, p.terms), "conv_split: unsupported geometry k=%dx%d stride %d with %d-term products", p.KH, p.KW, p.S, p.terms);
    // Split-K for grids with too few tiles for the chip (the coarse pyramid levels): decided from the per-image count of canonical
    // (4 rows x 32 px x 32 channels) tiles and the chunk count only, never from the batch or the tile shape -- the summation
    // order, hence the bits, of a pair must not depend on its batch mates.
    p.ksplit = 1;
    {
        const long blocks1 = (long)cdiv(p.Wo, 32) * cdiv(p.Ho, 4) * (p.cout_pad / 32);
        if (p.scratch && p.terms == 3 && p.S == 1 && blocks1 <= 256 && p.nchunk >= 2) {
            p.ksplit = (int)std::min<long>(std::min(8, p.nchunk), std::max<long>(1, 1024 / blocks1));
            if ((size_t)p.B * p.Ho * p.Wo * p.cout_pad * p.ksplit > p.scratch_floats) p.ksplit = 1;
        }
    }
    if (p.S == 2) return p.cout_pad % 64 == 0 ? launch_x<3, 1, 2, 5, 3, 2>(p, st) : launch_x<3, 1, 1, 5, 2, 2>(p, st);
    // small grids (three-term, 3 x 3 or smaller kernels): 4-row tiles while 8-row ones would leave CUs without work
    if (p.terms == 3 && p.KH <= 3 && p.KW <= 3) {
        const long t8 = (long)cdiv(p.Wo, 32) * cdiv(p.Ho, 8) * p.B * p.ksplit;
        if (p.cout_pad % 64 == 0 && t8 * (p.cout_pad / 64) < 512) return launch_x<3, 1, 2, 2, 3, 4>(p, st);
        if (p.cout_pad % 64 != 0 && t8 * (p.cout_pad / 32) < 512) return launch_x<3, 1, 1, 2, 2, 4>(p, st);
    }
    const bool wide = p.KW > 3;                       // 1 x k / k x k rows of 5 or 7 taps: the larger weight-row class
    const bool tall = (7 + p.KH) * (31 + p.KW) * 2 > 256 * 3;
    if (p.terms == 3) {
        // 128 channels, 3 x 3, at least two 16-row tiles per CU and six K chunks: the one-workgroup-per-CU kernel (bit-identical
        // results; 128->128 at 1024^2: 841 -> 769 us, at 512^2: 221 -> 196; fewer tiles or chunks: its prologue does not pay)
        const long t16 = (long)cdiv(p.Wo, 32) * cdiv(p.Ho, 16) * p.B;
        // 32-bit buffer offsets inside one image of every source (the 16-row kernel's descriptors and per-lane offsets are 32-bit;
        // computed from the real pixel strides: a wide caller tensor sends the layer to the small-tile kernel instead)
        bool fits31 = true;
        for (int i = 0; i < p.nseg; ++i)
            fits31 = fits31 && ((size_t)p.H * p.W * (size_t)p.seg[i].stride + (size_t)p.seg[i].cload) * 4 < ((size_t)1 << 31);
        // A folded tail (p.wtail) only exists for this kernel and sums in another order than the zero-padded tail chunk of the
        // others: for layers that have one, the kernel is chosen from the per-image tile count (a pair's bits must not depend on
        // its batch mates); layers without a tail give the same bits on every kernel and may follow the batch.
        ConvParamsX pt = p;
        const bool fold = p.wtail && !(PIV_KNOB(1) & 32);
        pt.tail = fold ? 1 : 0;
        const long t16b = fold ? (long)cdiv(p.Wo, 32) * cdiv(p.Ho, 16) : t16;
        if (p.ksplit == 1 && p.cout_pad % 128 == 0 && p.KH == 3 && p.KW == 3 && t16b * (p.cout_pad / 128) >= 512 &&
            (p.nchunk >= 6 || (fold && p.nchunk >= 4)) && fits31 && !(PIV_KNOB(1) & 262144))
            return launch_xbig<4>(pt, st);
        if (p.ksplit == 1 && p.cout_pad % 64 == 0 && p.KH == 3 && p.KW == 3 && t16b * (p.cout_pad / 64) >= 512 && p.nchunk >= 6 && fits31 &&
            !(PIV_KNOB(1) & 131072))
            return launch_xbig<2>(pt, st);       // 128->64 at 1024^2: 476 -> 445 us
        // 128-channel tiles unless that leaves fewer than four workgroups per CU (tile shapes never change a result's bits)
        const long t128 = (long)cdiv(p.Wo, 32) * cdiv(p.Ho, 8) * p.B * (p.cout_pad / 128);
        if (p.cout_pad % 128 == 0 && !wide && !tall && t128 >= 1024) return launch_x<3, 2, 4, 3, 6, 2>(p, st);
        if (p.cout_pad % 64 == 0) {
            if (wide) return launch_x<3, 2, 2, 5, 7, 2>(p, st);
            // 3 x 3 with 64 channels: 16-row tiles (12 LDS operand reads per 24 MFMAs instead of 8 per 12) while four workgroups per
            // CU remain; 128->64 at 1024^2: 478 -> 450 us.  (32-channel layers lose with 16 rows: 157 -> 162 us; 1 x 1: 183 -> 200.)
            const long t16 = (long)cdiv(p.Wo, 32) * cdiv(p.Ho, 16) * p.B * (p.cout_pad / 64);
            if (p.KH == 3 && p.KW == 3 && t16 >= 1024 && !(PIV_KNOB(1) & 524288)) return launch_x<3, 4, 2, 5, 3, 2>(p, st);
            return tall ? launch_x<3, 2, 2, 5, 3, 2>(p, st) : launch_x<3, 2, 2, 3, 3, 3>(p, st);
        }
        if (wide) return launch_x<3, 2, 1, 5, 4, 2>(p, st);
        return tall ? launch_x<3, 2, 1, 5, 2, 2>(p, st) : launch_x<3, 2, 1, 3, 2, 3>(p, st);
    }
    if (p.cout_pad % 64 == 0) {
        if (wide) return launch_x<6, 2, 2, 5, 11, 2>(p, st);
        return tall ? launch_x<6, 2, 2, 5, 5, 2>(p, st) : launch_x<6, 2, 2, 3, 5, 2>(p, st);
    }
    if (wide) return launch_x<6, 2, 1, 5, 6, 2>(p, st);
    return tall ? launch_x<6, 2, 1, 5, 3, 2>(p, st) : launch_x<6, 2, 1, 3, 3, 2>(p, st);
}

static unsigned short f16_bits(float v);

// The 4-lane tail chunk of a 3 x 3 layer with its taps folded into K (conv_split_big_kernel): [step g][piece][k-half][cout_pad][8],
// element j of k-half kb = tap 4g + 2kb + (j >> 2), channel c_tail + (j & 3); same power-of-two scale as pack_conv_x.
void pack_conv_x_tail(const float *w, int cout, int cin, int c_first, int c_real, float scale_inv, std::vector<unsigned short> &pk)
{
    const int cp = (cout + 31) / 32 * 32;
    const double s = 1.0 / (double)scale_inv;
    pk.assign((size_t)3 * 2 * 2 * cp * 8, 0);
    for (int g = 0; g < 3; ++g)
        for (int kb = 0; kb < 2; ++kb)
            for (int j = 0; j < 8; ++j) {
                const int t = 4 * g + 2 * kb + (j >> 2), c = j & 3;
                if (t >= 9 || c >= c_real) continue;
                for (int n = 0; n < cout; ++n) {
                    const double bw = (double)w[((size_t)n * cin + c_first + c) * 9 + t] * s;
                    const float bh = (float)(_Float16)(float)bw;
                    const float bm = (float)(_Float16)(float)((bw - (double)bh) * 2048.0);
                    const float piece[2] = {bh, bm * 0x1p-11f};
                    for (int q = 0; q < 2; ++q)
                        pk[((((size_t)g * 2 + q) * 2 + kb) * cp + n) * 8 + j] = f16_bits(piece[q]);
                }
            }
}

static unsigned short f16_bits(float v)
{
    const _Float16 h = (_Float16)v;
    unsigned short b;
    memcpy(&b, &h, 2);
    return b;
}

// OIHW fp32 weights -> fp16 pieces [chunk][tap][piece][k-half][cout_pad][8]; chunk = 16 staged input channels of one source.
// *out_scale = 2^-k, the factor that undoes the power-of-two weight scale (max |w| 2^k in [2^13, 2^14)).
void pack_conv_x(const float *w, int cout, int cin, int taps, const int *creal, const int *cload, const int *coff, int nseg,
                 std::vector<unsigned short> &pk, int *nchunk_out, float *out_scale)
{
    const int cp = (cout + 31) / 32 * 32;
    float wmax = 0.f;
    for (size_t i = 0; i < (size_t)cout * cin * taps; ++i)
        if (std::isfinite(w[i])) wmax = std::fmax(wmax, std::fabs(w[i]));
    int e = 0;
    if (wmax > 0.f) {
        std::frexp(wmax, &e);        // wmax = f 2^e, f in [0.5, 1)  ->  wmax 2^(14-e) in [2^13, 2^14)
        e = 14 - e;
    }
    e = std::max(-100, std::min(100, e));
    const double s = std::ldexp(1.0, e);
    *out_scale = (float)std::ldexp(1.0, -e);
    int nchunk = 0;
    for (int sg = 0; sg < nseg; ++sg) nchunk += (cload[sg] + 15) / 16;
    pk.assign((size_t)nchunk * taps * 6 * cp * 8, 0);
    int chunk = 0, run = 0;
    for (int sg = 0; sg < nseg; ++sg) {
        const int off = coff[sg] >= 0 ? coff[sg] : run;
        for (int c0 = 0; c0 < cload[sg]; c0 += 16, ++chunk)
            for (int t = 0; t < taps; ++t)
                for (int kb = 0; kb < 2; ++kb)
                    for (int j = 0; j < 8; ++j) {
                        const int c = c0 + 8 * kb + j;
                        if (c >= creal[sg]) continue;
                        for (int n = 0; n < cout; ++n) {
                            const double bw = (double)w[((size_t)n * cin + off + c) * taps + t] * s;       // exact
                            const float bh = (float)(_Float16)(float)bw;
                            const double r1 = (bw - (double)bh) * 2048.0;                                   // exact
                            const float bm = (float)(_Float16)(float)r1;
                            const double r2 = (r1 - (double)bm) * 2048.0;                                   // exact
                            const float bl = (float)(_Float16)(float)r2;
                            const float piece[3] = {bh, bm * 0x1p-11f, bl * 0x1p-22f};
                            for (int q = 0; q < 3; ++q)
                                pk[(((((size_t)chunk * taps + t) * 3 + q) * 2 + kb) * cp + n) * 8 + j] = f16_bits(piece[q]);
                        }
                    }
        run += creal[sg];
    }
    *nchunk_out = nchunk;
}

}  // namespace pivlfn
