// Direct (im2col-free) NHWC fp32 convolution on the gfx950 fp32 matrix cores.
//
// Covers every dense conv on the LiteFlowNet path (reference: torch.nn.Conv2d call sites in
// /root/reference/src/models.py:70-106, 124, 154-163, 197-207, 229-272): 3x3 s1/s2, 1x1, 7x7, kxk flow
// heads, (kx1)/(1xk) separable pairs; bias, optional LeakyReLU(0.1) and optional residual fused in the
// epilogue; up to three input sources so the concatenations of src/models.py:216 and :280 never
// materialise.
//
// Formulation: implicit GEMM  D[pixel][cout] = sum_{tap, cin} X[pixel+tap][cin] * W[cout][cin][tap]
//   * one workgroup = 4 waves = a TH x 32 output-pixel tile (TH = 4*MT rows) x BN = 32*NT channels;
//   * K is walked in chunks of 8 input channels: the (TH*S+KH-1) x (32*S+KW-1) x 8 input patch is staged
//     ONCE per chunk into LDS and reused by all KH*KW taps (this is what makes it im2col-free), together
//     with the chunk's [tap][8][BN] weight slab;
//   * v_mfma_f32_32x32x2_f32: A = 32 consecutive output pixels of one row, B = 32 output channels.
//     Each lane fetches 4 consecutive k with one ds_read_b128 and feeds them to 4 MFMAs, so MFMA j of a
//     chunk contracts input channels {j, 4+j}: a permutation of the k order inside a chunk, exact fp32
//     fma chains (the matrix core rounds once per product like fmaf).
//   * LDS pixel pitch is 12 floats (8 + 4 pad): the 16-lane groups of ds_read_b128 hit 16 distinct
//     16-byte slots, conflict-free at stride 1.
//   * accumulator layout (32x32): lane&31 = channel, so every epilogue store instruction writes two
//     full 128-byte channel runs.
#include <algorithm>
#include "common.h"

namespace pivlfn {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

constexpr int PIXP = 12;
constexpr unsigned C2OOB = 0x80000000u;      // buffer-load offset beyond every descriptor: reads as zero

template <int MT, int NT>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BN = NT * 32;
    constexpr int TH = 4 * MT;
    const int taps = p.KH * p.KW;
    const int PH = (TH - 1) * p.S + p.KH;
    const int PW = 31 * p.S + p.KW;
    float *patch = smem;
    float *wts = smem + PH * PW * PIXP;

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

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    int abase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) abase[m] = ((wave * MT + m) * p.S * PW + row * p.S) * PIXP + hh * 4;
    const int bbase = (hh * BN + row) * 4;

    const int npix = PH * PW;
    const f32x4 *wsrc = reinterpret_cast<const f32x4 *>(p.wpk);
    int chunk = 0;
    for (int s = 0; s < p.nseg; ++s) {
        const float *sp = p.seg[s].ptr;
        const int scl = p.seg[s].cload, sst = p.seg[s].stride;
        for (int c0 = 0; c0 < scl; c0 += 8, ++chunk) {
            if (chunk) __syncthreads();
            for (int idx = tid; idx < npix * 2; idx += 256) {
                const int pix = idx >> 1, q = idx & 1;
                const int py = pix / PW, px = pix - py * PW;
                const int iy = iy0 + py, ix = ix0 + px, c = c0 + q * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && c < scl)
                    v = *reinterpret_cast<const f32x4 *>(sp + (size_t)((b * p.H + iy) * p.W + ix) * sst + c);
                *reinterpret_cast<f32x4 *>(patch + pix * PIXP + q * 4) = v;
            }
            const f32x4 *wc = wsrc + (size_t)chunk * taps * 2 * p.cout_pad + n0;
            for (int idx = tid; idx < taps * 2 * BN; idx += 256) {
                const int th = idx / BN, n = idx - th * BN;
                reinterpret_cast<f32x4 *>(wts)[idx] = wc[(size_t)th * p.cout_pad + n];
            }
            __syncthreads();
            int tap = 0;
            for (int ky = 0; ky < p.KH; ++ky) {
                for (int kx = 0; kx < p.KW; ++kx, ++tap) {
                    const int toff = (ky * PW + kx) * PIXP;
                    f32x4 a[MT], bq[NT];
#pragma unroll
                    for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const f32x4 *>(patch + abase[m] + toff);
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        bq[n] = *reinterpret_cast<const f32x4 *>(wts + tap * 2 * BN * 4 + bbase + n * 128);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int n = 0; n < NT; ++n)
                                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][j], bq[n][j], acc[m][n], 0, 0, 0);
                }
            }
        }
    }

    // epilogue: bias, residual, activation, masked NHWC store
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int ch = n0 + n * 32 + row;
        if (ch >= p.cout_store) continue;
        const float bias = p.bias[ch];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int oy = y0 + wave * MT + m;
            if (oy >= p.Ho) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ox = x0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (ox >= p.Wo) continue;
                const size_t pix = (size_t)(b * p.Ho + oy) * p.Wo + ox;
                float v = acc[m][n][r] + bias;
                if (p.res) v += p.res[pix * p.res_stride + ch];
                if (p.lrelu) v = lrelu01(v);
                p.out[pix * p.out_stride + ch] = v;
            }
        }
    }
}


// ---- v2: same tiling, plus register prefetch ------------------------------------------------------------------------
// The next K-chunk's input patch and weight slab are loaded global -> registers BEFORE the current chunk's MFMAs and
// written to LDS after them, so the global/L2 latency hides behind the matrix work even with one workgroup per CU
// (the small pyramid levels, where v1 spent >90% of its time waiting on exposed staging).
// PMAX / WMAX = compile-time bounds on the 16-byte loads per thread for the patch / the weight slab.
// (64-accumulator tiles with the small staging class fit 168 VGPRs: three workgroups per CU)
template <int MT, int NT, int PMAX, int WMAX>
__global__ __launch_bounds__(256, (MT * NT <= 4 && NT <= 2 && PMAX <= 5) ? 3 : 2) void conv_mfma2_kernel(const ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BN = NT * 32;
    constexpr int TH = 4 * MT;
    const int taps = p.KH * p.KW;
    const int PH = (TH - 1) * p.S + p.KH;
    const int PW = 31 * p.S + p.KW;
    float *patch = smem;
    float *wts = smem + PH * PW * PIXP;

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

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    int abase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) abase[m] = ((wave * MT + m) * p.S * PW + row * p.S) * PIXP + hh * 4;
    const int bbase = (hh * BN + row) * 4;

    // this thread's staging slots: patch slot i covers (pixel, quad) = (idx>>1, idx&1), idx = tid + 256*i
    const int npix2 = PH * PW * 2;
    const int nw4 = taps * 2 * BN;
    // Staging loads are buffer loads through per-source, per-image descriptors: a slot outside the image (the zero padding) or past
    // the patch carries an out-of-range offset and the hardware's range check returns zeros -- no divergent branch around any load
    // (the predicated form cost an exec-mask save / branch / restore per load and turned the counted waits into vmcnt(0): on the
    // short K loops of NetC's stride-2 layers a quarter of a workgroup's time went into issuing them).
    // The descriptors start at the first image row of this workgroup's patch (64-bit scalar arithmetic): the 32-bit per-lane
    // offsets only span the patch rows, so one image of a source may be of any size (round 3: < 2 GiB).
    const int row0 = min(max(iy0, 0), p.H - 1);
    unsigned ppix[PMAX];       // index of the pixel relative to (row0, 0) of its image, C2OOB = zero
    int plds[PMAX];            // LDS float offset
#pragma unroll
    for (int i = 0; i < PMAX; ++i) {
        const int idx = tid + 256 * i;
        const int pix = idx >> 1, q = idx & 1;
        const int py = pix / PW, px = pix - py * PW;
        const int iy = iy0 + py, ix = ix0 + px;
        const bool ok = idx < npix2 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        ppix[i] = ok ? (unsigned)((iy - row0) * p.W + ix) : C2OOB;      // relative to the descriptor's first row
        plds[i] = idx < npix2 ? pix * PIXP + q * 4 : -1;
    }
    const int q4 = (tid & 1) * 4;
    unsigned woff[WMAX];       // byte offset inside the chunk's slab, C2OOB = none
#pragma unroll
    for (int i = 0; i < WMAX; ++i) {
        const int idx = tid + 256 * i;
        const int th = idx / BN, n = idx - th * BN;
        woff[i] = idx < nw4 ? (unsigned)(th * p.cout_pad + n0 + n) * 16u : C2OOB;
    }
    const unsigned wchunk_bytes = (unsigned)taps * 2u * (unsigned)p.cout_pad * 16u;
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.wpk), 0, (unsigned)p.nchunk * wchunk_bytes, 0x00020000);

    f32x4 pr[PMAX], wr[WMAX];
    // this workgroup's K range: all chunks, or with split-K (gridDim.z > 1) an even share of the full chunks; the 4-channel
    // tail chunk, if any, goes to the last share
    const int nfull = p.nchunk - p.tail;
    const int nz = gridDim.z, kz = blockIdx.z;
    const int per = (nfull + nz - 1) / nz;
    const int kc0 = min(nfull, kz * per), kc1 = min(nfull, kc0 + per);
    const bool has_tail = p.tail && kz == nz - 1;
    const int kend = kc1 + (has_tail ? 1 : 0);
    int seg = 0, c0 = kc0 * 8;
    while (seg + 1 < p.nseg && c0 >= p.seg[seg].cload) { c0 -= (p.seg[seg].cload + 7) / 8 * 8; ++seg; }
    int scl = p.seg[seg].cload, sst = p.seg[seg].stride;
    const size_t img_px = (size_t)p.H * p.W;
    const size_t px_left = (size_t)(p.H - row0) * p.W - 1;          // pixels from the base to the last one of the image
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.seg[seg].ptr + ((size_t)b * img_px + (size_t)row0 * p.W) * sst), 0,
                                                                  (unsigned)min((px_left * sst + scl) * 4, (size_t)0x7fffffff), 0x00020000);
    unsigned pvo[PMAX];        // byte offset of the slot's 16 bytes inside the current source's image (pixel record + quad), C2OOB = zero
#pragma unroll
    for (int i = 0; i < PMAX; ++i) pvo[i] = ppix[i] != C2OOB ? ppix[i] * (unsigned)(sst * 4) + (unsigned)q4 * 4u : C2OOB;

#ifdef PIVLFN_STAMPS
#define CONV2_ABL(BIT) (PIV_DBG(p) & (BIT))
#else
#define CONV2_ABL(BIT) false
#endif
#define CONV2_LOAD(CH)                                                                            \
    do {                                                                                          \
        if (CONV2_ABL(2)) break;                                                                  \
        const unsigned back_ = (scl - c0 <= 4) ? (unsigned)q4 * 4u : 0u;      /* 4-channel tail: both quads fetch the same 16 bytes (an out-of-range offset stays out of range) */ \
        _Pragma("unroll") for (int i = 0; i < PMAX; ++i)                                          \
            pr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(pvo[i] - back_), c0 * 4, 0)); \
        _Pragma("unroll") for (int i = 0; i < WMAX; ++i)                                          \
            wr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, (int)woff[i], (int)((unsigned)(CH)*wchunk_bytes), 0)); \
    } while (0)

    // K loop.  A source whose channel count is 4 (mod 8) ends in a half chunk; the packer only allows that for the LAST
    // source, so the tail is peeled: the hot loop below stays branch-free, and the tail contracts its 4 channels with two
    // MFMAs per tap (k = {j, 2+j}; staged as [c0 c1 c2 c3 | c2 c3 0 0] so both lane halves read their pair at j = 0, 1).
    // phase stamps (tools/bench_ops.py conv_stamps --tune3 -1; only in a -DPIVLFN_STAMPS build: the accumulators cost
    // registers the shipped <2,4> and <2,2> tiles do not have)
#ifdef PIVLFN_STAMPS
    const bool stamp = p.stamps != nullptr && wave == 0 && blockIdx.y == 0 && blockIdx.z == 0;
    unsigned long long tk = 0, d_commit = 0, d_bar1 = 0, d_issue = 0, d_taps = 0, d_bar2 = 0, t_begin = 0;
#define STAMP(ACC)                                                                                \
    do {                                                                                          \
        if (stamp) {                                                                              \
            const unsigned long long now_ = __builtin_readcyclecounter();                         \
            ACC += now_ - tk;                                                                     \
            tk = now_;                                                                            \
        }                                                                                         \
    } while (0)
    if (stamp) t_begin = tk = __builtin_readcyclecounter();
#else
#define STAMP(ACC) do { } while (0)
#endif
    if (kc0 < kend) CONV2_LOAD(kc0);
    for (int chunk = kc0; chunk < kc1; ++chunk) {
        // registers -> LDS (waits for the loads of this chunk)
#pragma unroll
        for (int i = 0; i < PMAX; ++i)
            if (plds[i] >= 0) *reinterpret_cast<f32x4 *>(patch + plds[i]) = pr[i];
#pragma unroll
        for (int i = 0; i < WMAX; ++i)
            if (woff[i] != C2OOB) reinterpret_cast<f32x4 *>(wts)[tid + 256 * i] = wr[i];
        STAMP(d_commit);
        __syncthreads();
        STAMP(d_bar1);
        // advance to the next chunk's source and put its loads in flight
        if (chunk + 1 < kend) {
            c0 += 8;
            if (c0 >= scl) {
                ++seg;
                c0 = 0;
                scl = p.seg[seg].cload; sst = p.seg[seg].stride;
                rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.seg[seg].ptr + ((size_t)b * img_px + (size_t)row0 * p.W) * sst), 0,
                                                       (unsigned)min((px_left * sst + scl) * 4, (size_t)0x7fffffff), 0x00020000);
#pragma unroll
                for (int i = 0; i < PMAX; ++i) pvo[i] = ppix[i] != C2OOB ? ppix[i] * (unsigned)(sst * 4) + (unsigned)q4 * 4u : C2OOB;
            }
            CONV2_LOAD(chunk + 1);
        }
        STAMP(d_issue);
        int tap = 0;
        for (int ky = 0; ky < (CONV2_ABL(1) ? 0 : p.KH); ++ky) {
            for (int kx = 0; kx < p.KW; ++kx, ++tap) {
                const int toff = (ky * PW + kx) * PIXP;
                f32x4 a[MT], bq[NT];
#pragma unroll
                for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const f32x4 *>(patch + abase[m] + toff);
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    bq[n] = *reinterpret_cast<const f32x4 *>(wts + tap * 2 * BN * 4 + bbase + n * 128);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[n][j], a[m][j], acc[m][n], 0, 0, 0);   // A = channels, B = pixels
            }
        }
        STAMP(d_taps);
        __syncthreads();       // all waves done with this chunk's LDS image before it is overwritten
        STAMP(d_bar2);
    }
#ifdef PIVLFN_STAMPS
    unsigned long long t_loop_end = 0;
    if (stamp) t_loop_end = __builtin_readcyclecounter();
#endif
    if (has_tail) {
#pragma unroll
        for (int i = 0; i < PMAX; ++i)
            if (plds[i] >= 0) {
                f32x4 v = pr[i];
                if (q4) v = f32x4{v[2], v[3], 0.f, 0.f};
                *reinterpret_cast<f32x4 *>(patch + plds[i]) = v;
            }
#pragma unroll
        for (int i = 0; i < WMAX; ++i)
            if (woff[i] != C2OOB) reinterpret_cast<f32x4 *>(wts)[tid + 256 * i] = wr[i];
        __syncthreads();
        int tap = 0;
        for (int ky = 0; ky < p.KH; ++ky) {
            for (int kx = 0; kx < p.KW; ++kx, ++tap) {
                const int toff = (ky * PW + kx) * PIXP;
                f32x2 a[MT], bq[NT];
#pragma unroll
                for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const f32x2 *>(patch + abase[m] + toff);
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    bq[n] = *reinterpret_cast<const f32x2 *>(wts + tap * 2 * BN * 4 + bbase + n * 128);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[n][j], a[m][j], acc[m][n], 0, 0, 0);
            }
        }
    }
#undef CONV2_LOAD

    // Epilogue.  With A = weights and B = pixels the accumulator tile is D[channel][pixel]: lane&31 = pixel, and registers
    // 4g..4g+3 hold channels 8g + 4*hh + {0,1,2,3} -- four consecutive channels per lane, so bias / residual / output move
    // as 16-byte vectors: 32 stores per thread for a 64x128 wave tile instead of 128, and a quarter of the address math.
    if (nz > 1) {       // split-K: raw partial sums, every channel of the padded N block, to scratch[kz][pixel][cout_pad]
        const int ox = x0 + row;
        const size_t npix = (size_t)p.B * p.Ho * p.Wo;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int oy = y0 + wave * MT + m;
            if (oy >= p.Ho || ox >= p.Wo) continue;
            float *prow = p.scratch + ((size_t)kz * npix + (size_t)(b * p.Ho + oy) * p.Wo + ox) * p.cout_pad + n0;
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4 *>(prow + n * 32 + 8 * g + 4 * hh) =
                        f32x4{acc[m][n][4 * g + 0], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]};
        }
    } else {
        const int ox = x0 + row;
        const bool interior = x0 + 32 <= p.Wo && y0 + TH <= p.Ho && n0 + BN <= p.cout_store;   // workgroup-uniform fast path
        // all bias vectors of this lane first (the staging registers are dead by now): a load between two stores would wait for
        // the store in front of it -- vmcnt counts loads and stores in one order -- and serialise the whole epilogue
        f32x4 bias4[NT][4];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) bias4[n][g] = *reinterpret_cast<const f32x4 *>(p.bias + n0 + n * 32 + 8 * g + 4 * hh);   // bias is padded to cout_pad
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int oy = y0 + wave * MT + m;
            const bool pix_ok = interior || (oy < p.Ho && ox < p.Wo);
            const size_t pix = (size_t)(b * p.Ho + (oy < p.Ho ? oy : 0)) * p.Wo + (ox < p.Wo ? ox : 0);
            float *orow = p.out + pix * p.out_stride;
            const float *rrow = p.res ? p.res + pix * p.res_stride : nullptr;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = n0 + n * 32 + 8 * g + 4 * hh;
                    if (!pix_ok || (!interior && ch >= p.cout_store)) continue;
                    if (CONV2_ABL(4) && acc[m][n][0] != 12345.f) continue;
                    f32x4 v = {acc[m][n][4 * g + 0], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]};
                    v += bias4[n][g];
                    if (rrow) v += *reinterpret_cast<const f32x4 *>(rrow + ch);
                    if (p.lrelu) {
                        v[0] = lrelu01(v[0]); v[1] = lrelu01(v[1]); v[2] = lrelu01(v[2]); v[3] = lrelu01(v[3]);
                    }
                    *reinterpret_cast<f32x4 *>(orow + ch) = v;
                }
            }
        }
    }
#ifdef PIVLFN_STAMPS
    if (stamp && lane == 0) {
        const unsigned long long t_end = __builtin_readcyclecounter();
        unsigned long long *o = p.stamps + (size_t)blockIdx.x * 8;
        o[0] = d_commit; o[1] = d_bar1; o[2] = d_issue; o[3] = d_taps; o[4] = d_bar2;
        o[5] = t_end - t_loop_end;       // (tail chunk +) epilogue
        o[6] = t_end - t_begin;
        o[7] = t_begin;
    }
#endif
#undef STAMP
}

// Second pass of a split-K layer: out = act(bias + residual + sum_z scratch[z]), z ascending (deterministic), 4 channels per thread.
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvParams p, int nz)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int cq = p.cout_pad / 4;
    const long npix = (long)p.B * p.Ho * p.Wo;
    if (idx >= npix * cq) return;
    const long pix = idx / cq;
    const int ch = (int)(idx - pix * cq) * 4;
    if (ch >= p.cout_store) return;
    f32x4 v = *reinterpret_cast<const f32x4 *>(p.scratch + (size_t)pix * p.cout_pad + ch);
    for (int z = 1; z < nz; ++z) v += *reinterpret_cast<const f32x4 *>(p.scratch + ((size_t)z * npix + pix) * p.cout_pad + ch);
    v += *reinterpret_cast<const f32x4 *>(p.bias + ch);
    if (p.res) v += *reinterpret_cast<const f32x4 *>(p.res + (size_t)pix * p.res_stride + ch);
    if (p.lrelu) {
        v[0] = lrelu01(v[0]); v[1] = lrelu01(v[1]); v[2] = lrelu01(v[2]); v[3] = lrelu01(v[3]);
    }
    *reinterpret_cast<f32x4 *>(p.out + (size_t)pix * p.out_stride + ch) = v;
}

int launch_splitk_reduce(const ConvParams &p, int nz, hipStream_t st)
{
    const long items = (long)p.B * p.Ho * p.Wo * (p.cout_pad / 4);
    hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, p, nz);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

template <int MT, int NT, int PMAX, int WMAX>
static int launch_t2(const ConvParams &p, hipStream_t st)
{
    constexpr int TH = 4 * MT, BN = NT * 32;
    const int PH = (TH - 1) * p.S + p.KH, PW = 31 * p.S + p.KW;
    const size_t lds = ((size_t)PH * PW * PIXP + (size_t)p.KH * p.KW * 2 * BN * 4) * sizeof(float);
    PIV_REQUIRE(lds <= 160 * 1024, "conv: LDS tile of %zu bytes exceeds 160 KiB (k=%dx%d s=%d)", lds, p.KH, p.KW, p.S);
    PIV_REQUIRE(PH * PW * 2 <= 256 * PMAX && p.KH * p.KW * 2 * BN <= 256 * WMAX, "conv: internal staging bound exceeded");
    static LdsAttr attr;
    if (int rc = ensure_dyn_lds(attr, reinterpret_cast<const void *>(conv_mfma2_kernel<MT, NT, PMAX, WMAX>), 160 * 1024)) return rc;
    const int tiles = cdiv(p.Wo, 32) * cdiv(p.Ho, TH) * p.B;
    const int nz = p.ksplit > 1 ? p.ksplit : 1;
    dim3 grid(tiles, p.cout_pad / BN, nz);
    hipLaunchKernelGGL((conv_mfma2_kernel<MT, NT, PMAX, WMAX>), grid, dim3(256), lds, st, p);
    PIV_CHECK_HIP(hipGetLastError());
    if (nz > 1) {
        const long items = (long)p.B * p.Ho * p.Wo * (p.cout_pad / 4);
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, p, nz);
        PIV_CHECK_HIP(hipGetLastError());
    }
    return PIVLFN_OK;
}

// ---- single-chunk layers with many taps (NetC.conv1: 7x7, 3 -> 32): weights resident, persistent over tiles ------------------
// With K = one 4-channel tail chunk the v2 kernel restages the whole weight slab (49 taps x 32 channels x 32 bytes = 50 KB)
// for every 8x32-pixel tile against 8.5 KB of input patch: the layer runs at the CU's load-path limit (57 TFLOP/s), not at the
// matrix pipe's.  Here a workgroup stages the slab once and walks tiles blockIdx.x, +gridDim.x, ...; the next tile's patch
// is prefetched into registers under the current tile's 196 MFMAs per wave.  Same arithmetic and k order as v2's tail path
// (patch staged as [c0 c1 c2 c3 | c2 c3 0 0], two MFMAs per tap), so the results are bit-identical.
template <int PMAX>
__global__ __launch_bounds__(256, 2) void conv_k1_kernel(const ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int MT = 2, TH = 8, BN = 32;
    const int taps = p.KH * p.KW;
    const int PH = (TH - 1) * p.S + p.KH, PW = 31 * p.S + p.KW;
    float *patch = smem;
    float *wts = smem + PH * PW * PIXP;
    const int tiles_x = (p.Wo + 31) >> 5, tiles_y = (p.Ho + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * p.B;
    const int n0 = blockIdx.y * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = lane & 31, hh = lane >> 5;
    if ((int)blockIdx.x >= ntiles) return;

    // weight slab -> LDS, once
    {
        const f32x4 *wsrc = reinterpret_cast<const f32x4 *>(p.wpk) + n0;
        for (int idx = tid; idx < taps * 2 * BN; idx += 256)
            reinterpret_cast<f32x4 *>(wts)[idx] = wsrc[(idx / BN) * p.cout_pad + (idx % BN)];
    }
    int abase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) abase[m] = ((wave * MT + m) * p.S * PW + row * p.S) * PIXP + hh * 4;
    const int bbase = (hh * BN + row) * 4;
    const int npix2 = PH * PW * 2;
    const int q4 = (tid & 1) * 4;
    const float *sp = p.seg[0].ptr;
    const int sst = p.seg[0].stride;
    const f32x4 bias4[4] = {*reinterpret_cast<const f32x4 *>(p.bias + n0 + 4 * hh), *reinterpret_cast<const f32x4 *>(p.bias + n0 + 8 + 4 * hh),
                            *reinterpret_cast<const f32x4 *>(p.bias + n0 + 16 + 4 * hh), *reinterpret_cast<const f32x4 *>(p.bias + n0 + 24 + 4 * hh)};

    f32x4 pr[PMAX];
#define K1_LOAD(T)                                                                                \
    do {                                                                                          \
        int t_ = (T);                                                                             \
        const int tx_ = t_ % tiles_x;                                                             \
        t_ /= tiles_x;                                                                            \
        const int b_ = t_ / tiles_y;                                                              \
        const int ix0_ = tx_ * 32 * p.S - p.padX, iy0_ = (t_ - b_ * tiles_y) * TH * p.S - p.padY; \
        _Pragma("unroll") for (int i = 0; i < PMAX; ++i) {                                        \
            const int idx_ = tid + 256 * i, pix_ = idx_ >> 1;                                     \
            const int py_ = pix_ / PW, px_ = pix_ - py_ * PW;                                     \
            const int iy_ = iy0_ + py_, ix_ = ix0_ + px_;                                         \
            f32x4 v = {0.f, 0.f, 0.f, 0.f};                                                       \
            if (idx_ < npix2 && iy_ >= 0 && iy_ < p.H && ix_ >= 0 && ix_ < p.W)                   \
                v = *reinterpret_cast<const f32x4 *>(sp + (size_t)((b_ * p.H + iy_) * p.W + ix_) * sst); \
            pr[i] = v;                                                                            \
        }                                                                                         \
    } while (0)

    K1_LOAD(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();          // previous tile's operand reads are done (first pass: nothing to wait for)
#pragma unroll
        for (int i = 0; i < PMAX; ++i) {
            const int idx = tid + 256 * i;
            if (idx < npix2) {
                f32x4 v = pr[i];
                if (q4) v = f32x4{v[2], v[3], 0.f, 0.f};
                *reinterpret_cast<f32x4 *>(patch + (idx >> 1) * PIXP + q4) = v;
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) K1_LOAD(tile + gridDim.x);
        f32x16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        int tap = 0;
        for (int ky = 0; ky < p.KH; ++ky)
            for (int kx = 0; kx < p.KW; ++kx, ++tap) {
                const int toff = (ky * PW + kx) * PIXP;
                f32x2 a[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const f32x2 *>(patch + abase[m] + toff);
                const f32x2 bq = *reinterpret_cast<const f32x2 *>(wts + tap * 2 * BN * 4 + bbase);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[j], a[m][j], acc[m], 0, 0, 0);
            }
        // epilogue (same lane layout as v2): 4 consecutive channels per lane and register group
        int t_ = tile;
        const int tx = t_ % tiles_x;
        t_ /= tiles_x;
        const int b = t_ / tiles_y;
        const int x0 = tx * 32, y0 = (t_ - b * tiles_y) * TH;
        const int ox = x0 + row;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int oy = y0 + wave * MT + m;
            if (oy >= p.Ho || ox >= p.Wo) continue;
            float *orow = p.out + ((size_t)(b * p.Ho + oy) * p.Wo + ox) * p.out_stride + n0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (n0 + 8 * g + 4 * hh >= p.cout_store) continue;
                f32x4 v = {acc[m][4 * g + 0], acc[m][4 * g + 1], acc[m][4 * g + 2], acc[m][4 * g + 3]};
                v += bias4[g];
                if (p.lrelu) {
                    v[0] = lrelu01(v[0]); v[1] = lrelu01(v[1]); v[2] = lrelu01(v[2]); v[3] = lrelu01(v[3]);
                }
                *reinterpret_cast<f32x4 *>(orow + 8 * g + 4 * hh) = v;
            }
        }
    }
#undef K1_LOAD
}

// Applies to: one K chunk that is a 4-channel tail, >= 16 taps, no residual, 32-channel output blocks, enough tiles.
static int launch_conv_k1(const ConvParams &p, hipStream_t st)
{
    if (p.nchunk != 1 || !p.tail || p.res || p.KH * p.KW < 16 || p.nseg != 1 || (PIV_KNOB(1) & 256)) return -1;
    const int PH = 7 * p.S + p.KH, PW = 31 * p.S + p.KW;
    if (PH * PW * 2 > 256 * 5) return -1;
    const size_t lds = ((size_t)PH * PW * PIXP + (size_t)p.KH * p.KW * 2 * 32 * 4) * sizeof(float);
    if (lds > 80 * 1024) return -1;
    const int tiles = cdiv(p.Wo, 32) * cdiv(p.Ho, 8) * p.B;
    if (tiles < 1024) return -1;
    static LdsAttr attr;
    if (int rc = ensure_dyn_lds(attr, reinterpret_cast<const void *>(conv_k1_kernel<5>), 80 * 1024)) return rc;
    const int nby = p.cout_pad / 32;
    hipLaunchKernelGGL((conv_k1_kernel<5>), dim3(std::max(1, 512 / nby), nby), dim3(256), lds, st, p);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

// ---- NetC.conv1 (7 x 7, 3 -> 32) with the taps packed into K -----------------------------------------------------------------
// conv_k1 contracts the 4-lane padded pixel per tap: two MFMAs (k = 2 each) of which the second carries one real channel -- 98
// MFMAs per 32 pixels for 147 real products per output.  Here the patch is stored in LDS with 3 floats per pixel, so the 21
// (tap, channel) slots of one kernel row are 21 consecutive floats: K runs over (row ky, slot pair j) and a lane half reads slot
// 2 j + hh with one immediate offset -- 7 x 11 = 77 MFMAs (the 22nd slot of a row reads the next pixel's first channel against a zero
// weight).  The 77 weight fragments stay in registers for the life of the persistent workgroup (no LDS reads for them), the patch is
// 6.5 KB: three workgroups per CU.  K order: rows, then slot pairs (conv_k1: taps, then channel pairs).
constexpr int C3_PW = 38, C3_PITCH = 116, C3_PH = 14;         // patch of an 8 x 32 tile, floats per patch row (38 x 3 + the pad slot)

// FUSE: the level-1 1 x 1 layers on top (Conv1Fuse): the accumulator layout of the 32 x 32 x 2 instruction is already its B-operand
// layout -- lane (pixel, hh) holds channels 8 g + 4 hh + e in register 4 g + e, and k-slot hh of the MFMA (g, e) wants exactly that
// channel -- so the activated accumulators feed the next MFMAs without a transpose: 16 MFMAs per 32 output channels.
// Round 6, where the fused kernel's 450 us go (tools/c3k7_ablate.sh, profiles/r06_c3k7_ablation.log): stores alone 248 us (1.34 GB at
// 5.4 TB/s), matrix work + loads alone 362 us, loads alone 95 us -- the phases overlap only partly.  Two ways of overlapping them more
// were built and are SLOWER on the same box: the next patch written to a second LDS buffer before the tile's stores are issued (so that
// no wait has stores in front of it; 470 vs 448 us), and the 1 x 1 blocks software-pipelined over two accumulator sets (MFMAs of block
// k + 1 between the activation / stores of block k; 522 vs 462 us; tools/kernels/conv_c3k7_pipelined_blocks.hip.txt).
template <bool FUSE, int ABL = 0>
__global__ __launch_bounds__(256, FUSE ? 2 : 3) void conv_c3k7_kernel(const ConvParams p, const Conv1Fuse f)
{
    __shared__ __attribute__((aligned(16))) float patch[C3_PH * C3_PITCH + 4];
    __shared__ __attribute__((aligned(16))) float w11s[FUSE ? 6 * 4 * 64 * 4 : 4];
    if (FUSE) {
        for (int i = threadIdx.x; i < 6 * 4 * 64; i += 256) reinterpret_cast<f32x4 *>(w11s)[i] = reinterpret_cast<const f32x4 *>(f.w11)[i];
    }
    constexpr int TH = 8;
    const int tiles_x = (p.Wo + 31) >> 5, tiles_y = (p.Ho + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * p.B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = lane & 31, hh = lane >> 5;
    if ((int)blockIdx.x >= ntiles) return;
    // weight fragments: slot s = 3 kx + c of row ky is wpk[((ky * 7 + kx) * 2 + (c >> 1)) * cout_pad + n][c & 1] (net.hip pack_conv,
    // 4-channel tail layout); slot 21 is the pad
    float wq[7][11];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int j = 0; j < 11; ++j) {
            const int s_ = 2 * j + hh;
            const int kx = s_ / 3, c = s_ - 3 * kx;
            wq[ky][j] = s_ < 21 ? p.wpk[(((size_t)(ky * 7 + kx) * 2 + (c >> 1)) * p.cout_pad + row) * 4 + (c & 1)] : 0.f;
        }
    for (int i = tid; i < C3_PH; i += 256) { patch[i * C3_PITCH + 114] = 0.f; patch[i * C3_PITCH + 115] = 0.f; }     // pad slots: finite
    const f32x4 bias4[4] = {*reinterpret_cast<const f32x4 *>(p.bias + 4 * hh), *reinterpret_cast<const f32x4 *>(p.bias + 8 + 4 * hh),
                            *reinterpret_cast<const f32x4 *>(p.bias + 16 + 4 * hh), *reinterpret_cast<const f32x4 *>(p.bias + 24 + 4 * hh)};
    const float *sp = p.seg[0].ptr;
    const int sst = p.seg[0].stride;
    int abase[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) abase[m] = (wave * 2 + m) * C3_PITCH + row * 3 + hh;
    f32x4 pr[3];
#define C3_LOAD(T)                                                                                \
    do {                                                                                          \
        int t_ = (T);                                                                             \
        const int tx_ = t_ % tiles_x;                                                             \
        t_ /= tiles_x;                                                                            \
        const int b_ = t_ / tiles_y;                                                              \
        const int ix0_ = tx_ * 32 - 3, iy0_ = (t_ - b_ * tiles_y) * TH - 3;                       \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                           \
            const int pix_ = tid + 256 * i;                                                       \
            const int py_ = pix_ / C3_PW, px_ = pix_ - py_ * C3_PW;                               \
            const int iy_ = iy0_ + py_, ix_ = ix0_ + px_;                                         \
            f32x4 v = {0.f, 0.f, 0.f, 0.f};                                                       \
            if (pix_ < C3_PH * C3_PW && iy_ >= 0 && iy_ < p.H && ix_ >= 0 && ix_ < p.W)           \
                v = *reinterpret_cast<const f32x4 *>(sp + ((size_t)(b_ * p.H + iy_) * p.W + ix_) * sst); \
            pr[i] = v;                                                                            \
        }                                                                                         \
    } while (0)

    // ABL != 0: instances of the tools build for timing runs (tools/c3k7_ablate.sh): 1 no stores of the fused 1 x 1 layers, 2 no conv1
    // stores, 4 no MFMAs of the fused layers, 8 no conv1 MFMAs, 16 no patch loads after the first
    constexpr int abl = ABL;
    C3_LOAD(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();          // the previous tile's operand reads are done
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int pix = tid + 256 * i;
            if (pix < C3_PH * C3_PW) {
                const int py = pix / C3_PW, px = pix - py * C3_PW;
                float *d = patch + py * C3_PITCH + px * 3;
                d[0] = pr[i][0]; d[1] = pr[i][1]; d[2] = pr[i][2];
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles && !(abl & 16)) C3_LOAD(tile + gridDim.x);
        f32x16 acc[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int j = 0; j < 11; ++j) {
                float a[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) a[m] = patch[abase[m] + ky * C3_PITCH + 2 * j];
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    if (!(abl & 8)) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[ky][j], a[m], acc[m], 0, 0, 0);
                    else acc[m][j & 15] += a[m];
            }
        int t_ = tile;
        const int tx = t_ % tiles_x;
        t_ /= tiles_x;
        const int b = t_ / tiles_y;
        const int x0 = tx * 32, y0 = (t_ - b * tiles_y) * TH;
        const int ox = x0 + row;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int oy = y0 + wave * 2 + m;
            const bool ok = oy < p.Ho && ox < p.Wo;
            float *orow = p.out + ((size_t)(b * p.Ho + (ok ? oy : 0)) * p.Wo + (ok ? ox : 0)) * p.out_stride;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {acc[m][4 * g + 0], acc[m][4 * g + 1], acc[m][4 * g + 2], acc[m][4 * g + 3]};
                v += bias4[g];
                if (p.lrelu) {
                    v[0] = lrelu01(v[0]); v[1] = lrelu01(v[1]); v[2] = lrelu01(v[2]); v[3] = lrelu01(v[3]);
                }
                if (ok && 8 * g + 4 * hh < p.cout_store && !(abl & 2)) *reinterpret_cast<f32x4 *>(orow + 8 * g + 4 * hh) = v;
                if (FUSE) { acc[m][4 * g + 0] = v[0]; acc[m][4 * g + 1] = v[1]; acc[m][4 * g + 2] = v[2]; acc[m][4 * g + 3] = v[3]; }
            }
        }
        if (FUSE) {
            // blocks 0, 1: NetC_ext (64 channels, every image); blocks 2..5: moduleFeat (128 channels, the first B_feat images)
            const int nblk = b < f.B_feat ? 6 : 2;
#pragma unroll 1
            for (int blk = 0; blk < nblk; ++blk) {
                f32x16 a2[2];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) a2[m][r] = 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 wv = reinterpret_cast<const f32x4 *>(w11s)[(blk * 4 + g) * 64 + lane];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int m = 0; m < 2; ++m)
                            if (!(abl & 4)) a2[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[e], acc[m][4 * g + e], a2[m], 0, 0, 0);
                            else a2[m][4 * g + e] += wv[e] * acc[m][4 * g + e];
                }
                const bool ext = blk < 2;
                const int cb = ext ? 32 * blk : 32 * (blk - 2), cs = ext ? 64 : 128;
                const float *bsrc = f.b11 + (ext ? 0 : 64) + cb + 4 * hh;
                float *obase = ext ? f.out_ext : f.out_feat;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int oy = y0 + wave * 2 + m;
                    if (oy >= p.Ho || ox >= p.Wo) continue;
                    float *orow = obase + ((size_t)(b * p.Ho + oy) * p.Wo + ox) * cs + cb + 4 * hh;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v = {a2[m][4 * g + 0], a2[m][4 * g + 1], a2[m][4 * g + 2], a2[m][4 * g + 3]};
                        v += *reinterpret_cast<const f32x4 *>(bsrc + 8 * g);
                        v[0] = lrelu01(v[0]); v[1] = lrelu01(v[1]); v[2] = lrelu01(v[2]); v[3] = lrelu01(v[3]);
                        if (!(abl & 1) || v[0] == 1.2345e38f) *reinterpret_cast<f32x4 *>(orow + 8 * g) = v;
                    }
                }
            }
        }
    }
#undef C3_LOAD
}

// Applies to: 7 x 7, stride 1, pad 3, one source of 3 real channels on 4 lanes, 32 output channels, no residual, >= 1024 tiles
static int launch_conv_c3k7(const ConvParams &p, hipStream_t st)
{
    if (p.KH != 7 || p.KW != 7 || p.S != 1 || p.padY != 3 || p.padX != 3 || p.nseg != 1 || p.seg[0].cload != 4 || p.nchunk != 1 || !p.tail ||
        p.res || p.cout_pad != 32 || p.cin_real != 3 || (PIV_KNOB(1) & 134217728))
        return -1;
    const long tiles = (long)cdiv(p.Wo, 32) * cdiv(p.Ho, 8) * p.B;
    if ((long)cdiv(p.Wo, 32) * cdiv(p.Ho, 8) < 512) return -1;             // per image: never a function of the batch
    const int cus = device_cus();
    hipLaunchKernelGGL((conv_c3k7_kernel<false>), dim3((unsigned)std::min<long>(tiles, 3L * cus)), dim3(256), 0, st, p, Conv1Fuse{});
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

// conv1 + NetC_ext + moduleFeat of level 1 in one launch (same applicability as launch_conv_c3k7; cout_store 32, LeakyReLU on)
int launch_conv1_fused(const ConvParams &p, const Conv1Fuse &f, hipStream_t st)
{
    if (p.KH != 7 || p.KW != 7 || p.S != 1 || p.padY != 3 || p.padX != 3 || p.nseg != 1 || p.seg[0].cload != 4 || p.nchunk != 1 || !p.tail ||
        p.res || p.cout_pad != 32 || p.cin_real != 3 || p.cout_store != 32 || !p.lrelu || !f.w11 || !f.b11 || !f.out_ext || !f.out_feat)
        return -1;
    if ((long)cdiv(p.Wo, 32) * cdiv(p.Ho, 8) < 512) return -1;             // per image: never a function of the batch
    const long tiles = (long)cdiv(p.Wo, 32) * cdiv(p.Ho, 8) * p.B;
    const int cus = device_cus();
#ifdef PIVLFN_TOOLS
#define C3_ABL(M) case M: hipLaunchKernelGGL((conv_c3k7_kernel<true, M>), dim3((unsigned)std::min<long>(tiles, 2L * cus)), dim3(256), 0, st, p, f); break;
    switch (PIV_KNOB(7)) {
        C3_ABL(1) C3_ABL(2) C3_ABL(3) C3_ABL(4) C3_ABL(5) C3_ABL(12) C3_ABL(15) C3_ABL(28) C3_ABL(31)
        default: hipLaunchKernelGGL((conv_c3k7_kernel<true>), dim3((unsigned)std::min<long>(tiles, 2L * cus)), dim3(256), 0, st, p, f);
    }
#undef C3_ABL
#else
    hipLaunchKernelGGL((conv_c3k7_kernel<true>), dim3((unsigned)std::min<long>(tiles, 2L * cus)), dim3(256), 0, st, p, f);
#endif
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

// ---- 3 x 3 / stride 2 from 32 channels (NetC.conv2.0 32 -> 32 at 1024^2, NetC.conv3.0 32 -> 64 at 512^2) -------------------------
// The v2 kernel stages 8 channels of its patch per K chunk: 32 of a pixel's 128 bytes, four times over a workgroup's life, and with
// ~100 patches of 90-270 KB in flight per XCD the line has left that L2 before the next chunk asks for it -- the layer fetched its
// input 3.3 times (FETCH_SIZE, round 3) and ran at the HBM bound of that traffic.  Here a pixel is fetched once, as one whole line:
// the patch holds all 32 channels (17 x 33 pixels for an 8 x 16 output tile: 2 rows x 16 columns per wave), the weights of the whole
// layer stay in LDS for the workgroup's life, a workgroup walks tiles blockIdx.x, + gridDim.x, ... with the next tile's patch
// prefetched into registers under the current tile's MFMAs (one workgroup per CU: the patch and the weights take 118-155 KB).
// K order per output: the four 8-channel groups outer, taps inner, as in the v2 kernel.
// (Measured and dropped: the weight fragments straight from global memory, ten steps ahead of their use, so that the patch alone
// is in LDS and two workgroups fit a CU -- 291 us against 119 on 32 -> 32 at 1024^2: the fragment loads queue behind the next
// patch's HBM loads in the in-order wait counter, and 36 KB of fragments per tile and wave do not stay in L1.)
constexpr int S2_PIXP = 36, S2_PH = 17, S2_PW = 33, S2_NPIX = S2_PH * S2_PW, S2_PMAX = 18;     // 561 x 8 quads <= 256 x 18

template <int NT>
__global__ __launch_bounds__(256) void conv_s2c32_kernel(const ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BN = 32 * NT;
    float *patch = smem;
    float *wts = smem + S2_NPIX * S2_PIXP;
    const int tiles_x = (p.Wo + 15) >> 4, tiles_y = (p.Ho + 7) >> 3;
    const int ntiles = tiles_x * tiles_y * p.B;
    const int n0 = blockIdx.y * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = lane & 31, hh = lane >> 5, prow = (lane >> 4) & 1, pcol = lane & 15;
    if ((int)blockIdx.x >= ntiles) return;
    {   // the layer's weights for this block of output channels -> LDS, once: [group][tap][half][BN][4]
        const f32x4 *wsrc = reinterpret_cast<const f32x4 *>(p.wpk) + n0;
        for (int idx = tid; idx < 4 * 9 * 2 * BN; idx += 256)
            reinterpret_cast<f32x4 *>(wts)[idx] = wsrc[(idx / BN) * p.cout_pad + (idx % BN)];
    }
    const int abase = ((2 * (wave * 2 + prow)) * S2_PW + 2 * pcol) * S2_PIXP + hh * 4;
    const int bbase = (hh * BN + row) * 4;
    const float *sp = p.seg[0].ptr;
    const int sst = p.seg[0].stride;
    const int q = tid & 7, pix0 = tid >> 3;             // staging slot i: quad q of patch pixel pix0 + 32 i
    f32x4 bias4[NT][4];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) bias4[n][g] = *reinterpret_cast<const f32x4 *>(p.bias + n0 + n * 32 + 8 * g + 4 * hh);
    // patch coordinates of this thread's slots (the same for every tile): py << 8 | px, 0xffff = past the patch
    unsigned pyx[S2_PMAX];
#pragma unroll
    for (int i = 0; i < S2_PMAX; ++i) {
        const int pix = pix0 + 32 * i;
        const int py = pix / S2_PW, px = pix - py * S2_PW;
        pyx[i] = pix < S2_NPIX ? (unsigned)(py << 8 | px) : 0xffffu;
    }

    f32x4 pr[S2_PMAX];
    // Buffer loads through a descriptor that starts at the first image row of the tile's patch: zero padding and slots past the patch
    // carry an out-of-range offset (no branch around a load), and one image may be of any size.
#define S2_LOAD(T)                                                                                \
    do {                                                                                          \
        int t_ = (T);                                                                             \
        const int tx_ = t_ % tiles_x;                                                             \
        t_ /= tiles_x;                                                                            \
        const int b_ = t_ / tiles_y;                                                              \
        const int ix0_ = tx_ * 32 - 1, iy0_ = (t_ - b_ * tiles_y) * 16 - 1;                       \
        const int row0_ = min(max(iy0_, 0), p.H - 1);                                             \
        const size_t left_ = (size_t)(p.H - row0_) * p.W - 1;                                     \
        const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                     \
            const_cast<float *>(sp + ((size_t)b_ * p.H * p.W + (size_t)row0_ * p.W) * sst), 0,    \
            (unsigned)min((left_ * sst + 32) * 4, (size_t)0x7fffffff), 0x00020000);               \
        _Pragma("unroll") for (int i = 0; i < S2_PMAX; ++i) {                                     \
            const int iy_ = iy0_ + (int)(pyx[i] >> 8), ix_ = ix0_ + (int)(pyx[i] & 255u);         \
            const bool ok_ = (pyx[i] != 0xffffu) & (iy_ >= 0) & (iy_ < p.H) & (ix_ >= 0) & (ix_ < p.W); \
            const unsigned off_ = ok_ ? (unsigned)((iy_ - row0_) * p.W + ix_) * (unsigned)(sst * 4) + 16u * q : C2OOB; \
            pr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_, (int)off_, 0, 0)); \
        }                                                                                         \
    } while (0)

#ifndef S2_ABL_CT
#define S2_ABL_CT 0               // timing-only ablations (tools/ab_variant.sh): 1 no MFMAs, 2 no patch loads after the first, 4 no stores, 8 no LDS commit, 16 no barriers
#endif
    S2_LOAD(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (!(S2_ABL_CT & 16)) __syncthreads();          // the previous tile's operand reads are done (first pass: the weights are written)
#pragma unroll
        for (int i = 0; i < S2_PMAX; ++i)
            if (pyx[i] != 0xffffu && (!(S2_ABL_CT & 8) || tile == (int)blockIdx.x)) *reinterpret_cast<f32x4 *>(patch + (pix0 + 32 * i) * S2_PIXP + 4 * q) = pr[i];
        if (!(S2_ABL_CT & 16)) __syncthreads();
        if (tile + (int)gridDim.x < ntiles && !(S2_ABL_CT & 2)) S2_LOAD(tile + gridDim.x);
        f32x16 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
        // 36 steps (group c, tap): one 16-byte read of either operand feeds four MFMAs.  The reads run S2_D steps ahead of their
        // MFMAs, pinned by scheduling fences: with one wave per SIMD nothing else covers an LDS round trip (left to the compiler the
        // reads sit right in front of their use).  Measured: the same 120 us on 32 -> 32 at 1024^2 either way -- a tile takes
        // 17.8 k cycles for 9.2 k of matrix work.  Compile-time ablations (S2_ABL_CT, tools/ab_variant.sh; us per launch): whole
        // 121.7, without the MFMAs 70.9, without the next patch's loads 96.9, without the stores 109.0, without the LDS commit
        // 116.8, without the barriers 122.1, with none of the first four 22.5 -- the pieces ADD (51 + 25 + 13 + 5 + 22): one wave per
        // SIMD runs them in order and nothing else is there to overlap them.  Two workgroups per CU need the weights out of LDS
        // (patch 80.8 KB): from global memory that was 291 us (header), in registers 144 + 72 staging registers do not fit 256.
        constexpr int S2_D = 3;
        f32x4 av[36], bv[36][NT];
#define S2_READ(S)                                                                                \
        do {                                                                                      \
            const int c_ = (S) / 9, tap_ = (S) % 9, ky_ = tap_ / 3, kx_ = tap_ % 3;               \
            av[S] = *reinterpret_cast<const f32x4 *>(patch + abase + (ky_ * S2_PW + kx_) * S2_PIXP + 8 * c_); \
            _Pragma("unroll") for (int n = 0; n < NT; ++n)                                        \
                bv[S][n] = *reinterpret_cast<const f32x4 *>(wts + ((c_ * 9 + tap_) * 2 * BN) * 4 + bbase + n * 128); \
        } while (0)
#pragma unroll
        for (int s_ = 0; s_ < S2_D; ++s_) S2_READ(s_);
#pragma unroll
        for (int s_ = 0; s_ < 36; ++s_) {
            if (s_ + S2_D < 36) S2_READ(s_ + S2_D);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    if (S2_ABL_CT & 1) acc[n][(s_ + j) & 15] += bv[s_][n][j] * av[s_][j];
                    else acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[s_][n][j], av[s_][j], acc[n], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef S2_READ
        // epilogue: lane & 31 = pixel (2 rows x 16 columns of the wave), registers 4 g .. 4 g + 3 = channels 8 g + 4 hh + {0..3}
        int t_ = tile;
        const int tx = t_ % tiles_x;
        t_ /= tiles_x;
        const int b = t_ / tiles_y;
        const int oy = (t_ - b * tiles_y) * 8 + wave * 2 + prow, ox = tx * 16 + pcol;
        if (oy < p.Ho && ox < p.Wo && (!(S2_ABL_CT & 4) || acc[0][0] == 12345.f)) {
            float *orow = p.out + ((size_t)(b * p.Ho + oy) * p.Wo + ox) * p.out_stride + n0;
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (n0 + n * 32 + 8 * g + 4 * hh >= p.cout_store) continue;
                    f32x4 v = {acc[n][4 * g + 0], acc[n][4 * g + 1], acc[n][4 * g + 2], acc[n][4 * g + 3]};
                    v += bias4[n][g];
                    if (p.lrelu) {
                        v[0] = lrelu01(v[0]); v[1] = lrelu01(v[1]); v[2] = lrelu01(v[2]); v[3] = lrelu01(v[3]);
                    }
                    *reinterpret_cast<f32x4 *>(orow + n * 32 + 8 * g + 4 * hh) = v;
                }
        }
    }
#undef S2_LOAD
}

// Applies to: 3 x 3, stride 2, pad 1, one source of exactly 32 channels, 32 or 64 output channels, no residual, and (per image, never
// a function of the batch) at least 256 tiles of 8 x 16 outputs.
static int launch_conv_s2(const ConvParams &p_in, hipStream_t st)
{
    const ConvParams &p = p_in;
    if (p.KH != 3 || p.KW != 3 || p.S != 2 || p.padY != 1 || p.padX != 1 || p.nseg != 1 || p.seg[0].cload != 32 || p.nchunk != 4 || p.tail ||
        p.res || (p.cout_pad != 32 && p.cout_pad != 64) || (PIV_KNOB(1) & 4194304))
        return -1;
    if ((long)cdiv(p.Wo, 16) * cdiv(p.Ho, 8) < 256) return -1;
    if ((size_t)17 * p.W * p.seg[0].stride * 4 >= 0x7fffffffull) return -1;       // one patch inside a descriptor's 2 GiB
    const int nt = p.cout_pad / 32;
    const size_t lds = ((size_t)S2_NPIX * S2_PIXP + (size_t)4 * 9 * 2 * 32 * nt * 4) * sizeof(float);
    const int cus = device_cus();
    const long tiles = (long)cdiv(p.Wo, 16) * cdiv(p.Ho, 8) * p.B;
    const dim3 grid((unsigned)std::min<long>(tiles, cus), 1);
    static LdsAttr attr1, attr2;
    if (nt == 1) {
        if (int rc = ensure_dyn_lds(attr1, reinterpret_cast<const void *>(conv_s2c32_kernel<1>), 160 * 1024)) return rc;
        hipLaunchKernelGGL((conv_s2c32_kernel<1>), grid, dim3(256), lds, st, p);
    } else {
        if (int rc = ensure_dyn_lds(attr2, reinterpret_cast<const void *>(conv_s2c32_kernel<2>), 160 * 1024)) return rc;
        hipLaunchKernelGGL((conv_s2c32_kernel<2>), grid, dim3(256), lds, st, p);
    }
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

// (Measured and dropped in round 2: a dedicated streaming kernel for the HBM-bound 1x1 layers -- NetC_ext 32 -> 64, moduleFeat
// 32 -> 128 -- with the whole weight matrix resident in LDS, operands read straight from global memory one pixel run ahead and no
// barriers: 193.7 vs 195.8 us (moduleFeat, level 1) and 216.7 vs 235.3 us (NetC_ext) against this kernel's 16-row tiles, and no
// change of the step time.  Both run at ~3.5 TB/s of a 4:1 write-heavy stream; tools/bench_ops.py conv --filter 1x1.)

// Tile choice for v2.  Staging loads per thread: patch = PH*PW*2/256, slab = taps*2*BN/256 (16-byte each).
static int launch_conv2(const ConvParams &p_in, hipStream_t st)
{
    ConvParams p = p_in;
    p.stamps = reinterpret_cast<unsigned long long *>(((unsigned long long)(unsigned)PIV_KNOB(6) << 32) | (unsigned)PIV_KNOB(5));   // tools only
    PIV_SET_DBG(p, PIV_KNOB(7));
    const int taps = p.KH * p.KW;
    for (int s = 0; s < p.nseg; ++s)      // 32-bit byte offsets inside the rows of one patch (the descriptors are rebased per workgroup)
        PIV_REQUIRE((size_t)(15 * p.S + p.KH) * p.W * p.seg[s].stride * 4 < 0x7fffffffull, "conv: one patch (%d rows x %d x %d floats) of source %d exceeds the 2 GiB buffer-descriptor range", 15 * p.S + p.KH, p.W, p.seg[s].stride, s);
    // Split-K when one image has too few tiles for the chip and the K loop is long enough to be worth sharing.  Decided from
    // the per-image count of canonical (4 rows x 32 px x 32 channels) tiles only -- never from the batch size or from the tile
    // shape picked below (which does depend on it): a pair's flow must not depend on its batch mates, bit for bit.
    {
        const int nfull = p.nchunk - p.tail;
        const long blocks1 = (long)cdiv(p.Wo, 32) * cdiv(p.Ho, 4) * (p.cout_pad / 32);
        p.ksplit = 1;
        if (p.scratch && blocks1 <= 128 && nfull >= 4 && !(PIV_KNOB(1) & 128)) {
            p.ksplit = (int)std::min<long>(std::min(8, nfull / 2), std::max<long>(1, 512 / blocks1));
            if ((size_t)p.B * p.Ho * p.Wo * p.cout_pad * p.ksplit > p.scratch_floats) p.ksplit = 1;   // standalone layers with a small scratch only
        }
    }
    const long px_blocks1 = (long)cdiv(p.Wo, 32) * cdiv(p.Ho, 4) * p.B;     // workgroups with MT = 1 per N block
    // widest N tile (32*nt channels) that still leaves >= 256 workgroups; with fewer the tile is narrowed so more CUs get
    // work (the input patch is then re-staged once per N block, which is cheap at the small levels where this happens)
    int nt = 1;
    for (int cand = 4; cand >= 1; --cand) {
        if (p.cout_pad % (cand * 32)) continue;
        if (px_blocks1 * (p.cout_pad / (cand * 32)) >= 256 || cand == 1) { nt = cand; break; }
    }
    if ((PIV_KNOB(1) & 8) && nt == 4) nt = 2;              // A/B: 64-channel N tiles (three workgroups per CU) for 128-channel layers
    int mt = (px_blocks1 / 2) * (p.cout_pad / (nt * 32)) >= 512 ? 2 : 1;
    auto need = [&](int mt_, int nt_, int &pm, int &wm) {
        const int PH = (4 * mt_ - 1) * p.S + p.KH, PW = 31 * p.S + p.KW;
        pm = cdiv(PH * PW * 2, 256);
        wm = cdiv(taps * 2 * nt_ * 32, 256);
    };
    int pm, wm;
    // 16-row tiles for the 64-channel layers of the fine levels: the weight slab and the patch halo a workgroup restages per
    // K chunk are amortised over twice the pixels (same accumulator budget as the 8-row x 128-channel tile).  Measured at level 1:
    // 128->64 113 -> 129 TFLOP/s, 64->64 104 -> 119; the 32-channel layers LOSE with 16 rows (116 -> 96) and keep 8.
    if (nt == 2 && mt == 2 && (px_blocks1 / 4) * (p.cout_pad / 64) >= 512 && !(PIV_KNOB(1) & 16)) {
        need(4, nt, pm, wm);
        if (pm <= 5 && wm <= 5) return launch_t2<4, 2, 5, 5>(p, st);
    }
    need(mt, nt, pm, wm);
    if ((pm > 3 || wm > 9) && mt == 2 && nt >= 3) { mt = 1; need(mt, nt, pm, wm); }   // keep the big-staging class under 256 VGPRs
    if (pm > 9 || wm > 13) return -1;      // not covered: caller falls back to v1
    const bool small = pm <= 3 && wm <= 9;
    // a middle staging class (<= 5 patch loads, <= 5 slab loads per thread) for the k x 1 layers of conv_dist_R (7 x 1: a 14-row
    // patch, 4 + 4 loads): 168 instead of 240 registers, three workgroups per CU instead of two
    const bool mid = !small && pm <= 5 && wm <= 5 && !(PIV_KNOB(1) & 4);
#define PICK(MT_, NT_)                                                                     \
    (small ? launch_t2<MT_, NT_, 3, 9>(p, st) : launch_t2<MT_, NT_, 9, 13>(p, st))
    if (mt == 2) {
        switch (nt) {
            case 4: return small ? launch_t2<2, 4, 3, 9>(p, st) : -1;
            case 3: return small ? launch_t2<2, 3, 3, 9>(p, st) : -1;
            case 2: return mid ? launch_t2<2, 2, 5, 5>(p, st) : PICK(2, 2);
            default: return mid ? launch_t2<2, 1, 5, 5>(p, st) : PICK(2, 1);
        }
    }
    switch (nt) {
        case 4: return PICK(1, 4);
        case 3: return PICK(1, 3);
        case 2: return PICK(1, 2);
        default: return PICK(1, 1);
    }
#undef PICK
}


template <int MT, int NT>
static int launch_t(const ConvParams &p, hipStream_t st)
{
    constexpr int TH = 4 * MT, BN = NT * 32;
    const int PH = (TH - 1) * p.S + p.KH, PW = 31 * p.S + p.KW;
    const size_t lds = ((size_t)PH * PW * PIXP + (size_t)p.KH * p.KW * 2 * BN * 4) * sizeof(float);
    PIV_REQUIRE(lds <= 160 * 1024, "conv: LDS tile of %zu bytes exceeds 160 KiB (k=%dx%d s=%d)", lds, p.KH, p.KW, p.S);
    static LdsAttr attr;
    if (int rc = ensure_dyn_lds(attr, reinterpret_cast<const void *>(conv_mfma_kernel<MT, NT>), 160 * 1024)) return rc;
    const int tiles = cdiv(p.Wo, 32) * cdiv(p.Ho, TH) * p.B;
    dim3 grid(tiles, p.cout_pad / BN);
    hipLaunchKernelGGL((conv_mfma_kernel<MT, NT>), grid, dim3(256), lds, st, p);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

int launch_conv(const ConvParams &p, hipStream_t st)
{
    PIV_REQUIRE(p.nseg >= 1 && p.nseg <= 3, "conv: nseg=%d", p.nseg);
    PIV_REQUIRE(p.cout_pad % 32 == 0 && p.cout_store <= p.cout_pad, "conv: cout_pad=%d cout_store=%d", p.cout_pad, p.cout_store);
    PIV_REQUIRE(p.B > 0 && p.H > 0 && p.W > 0 && p.Ho > 0 && p.Wo > 0, "conv: empty shape");
    for (int s = 0; s < p.nseg; ++s)
        PIV_REQUIRE(p.seg[s].cload % 4 == 0 && p.seg[s].stride % 4 == 0 && p.seg[s].ptr, "conv: segment %d misaligned", s);
    if (!(PIV_KNOB(1) & 2)) {               // shipped path: v2 (register prefetch); knob bit 1 forces v1 for A/B
        const int rc3 = launch_conv_c3k7(p, st);
        if (rc3 >= 0) return rc3;
        const int rk = launch_conv_k1(p, st);
        if (rk >= 0) return rk;
        const int rs2 = launch_conv_s2(p, st);
        if (rs2 >= 0) return rs2;
        const int rc = launch_conv2(p, st);
        if (rc >= 0) return rc;
    }
    PIV_REQUIRE(!p.tail, "conv: this layer's weights are packed with a 4-channel tail chunk, which only the v2 kernel understands");
    int nt;
    if (p.cout_pad % 128 == 0) nt = 4;
    else if (p.cout_pad % 96 == 0) nt = 3;
    else if (p.cout_pad % 64 == 0) nt = 2;
    else nt = 1;
    // Two output rows per wave when that still leaves a full chip's worth of workgroups.
    const long blocks2 = (long)cdiv(p.Wo, 32) * cdiv(p.Ho, 8) * p.B * (p.cout_pad / (nt * 32));
    const bool big = blocks2 >= 512 && (p.KH * p.KW <= 25 || nt == 1);
    if (big) {
        switch (nt) {
            case 4: return launch_t<2, 4>(p, st);
            case 3: return launch_t<2, 3>(p, st);
            case 2: return launch_t<2, 2>(p, st);
            default: return launch_t<2, 1>(p, st);
        }
    }
    switch (nt) {
        case 4: return launch_t<1, 4>(p, st);
        case 3: return launch_t<1, 3>(p, st);
        case 2: return launch_t<1, 2>(p, st);
        default: return launch_t<1, 1>(p, st);
    }
}

}  // namespace pivlfn
