// Fused bilinear back-warp + 7x7 local cost volume (+ LeakyReLU) for gfx950.
//
// Replaces, in one launch and with no intermediate tensor:
//   backwarp(feat2, flow*scale)                      /root/reference/src/models.py:20-35, :171
//   kernel_Correlation_rearrange x2 + zero fills     /root/reference/src/correlation.py:9-34, 288-324
//   kernel_Correlation_updateOutput                  /root/reference/src/correlation.py:36-104, 326-337
//   leaky_relu(., 0.1)                               /root/reference/src/models.py:174-184
// out[b, 7(dy+3)+(dx+3), y, x] = (1/C) sum_c f1[b,c,s*y,s*x] * f2w[b,c,s*(y+dy),s*(x+dx)]  (zeros outside),
// f2w[b,c,Y,X] = bilinear(f2[b,c], X + scale*u[b,Y,X], Y + scale*v[b,Y,X])               (zeros outside).
//
// Structure (one workgroup = 256 threads = an 8x8 tile of output pixels):
//   * the warped second feature map is needed only at the (8+6)^2 = 196 stride-s positions around the tile;
//     those vectors are gathered (4 bilinear taps, channel-contiguous 16-byte loads in the NHWC layout),
//     blended and written ONCE into LDS, channels in chunks of CC; at stride 2 three quarters of the warp
//     of the reference never happens;
//   * lane = output pixel (64 per tile), wave = displacement group (d = wave, wave+4, ...): every lane
//     keeps its f1 vector in registers and streams the f2w vectors with ds_read_b128 (pixel pitch CC+4
//     floats -> conflict-free), 13 private accumulators, no cross-lane reduction at all;
//   * results are transposed through LDS so the [B,Ho,Wo,56] store is made of whole 16-byte lanes;
//   * workgroup ids are remapped so every XCD owns a contiguous band of tiles (halo re-reads hit its L2).
#include <cstdlib>
#include <hip/hip_ext.h>
#include <algorithm>
#include "common.h"

namespace pivlfn {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int TO = 8;
constexpr int TP = TO + 6;
constexpr int NPOS = TP * TP;
constexpr int OUTC = 56;

// Optional start/stop events attached to the next warp+correlation dispatch itself (hipExtLaunchKernelGGL): the
// same start/end timestamps rocprofv3 reports, with no marker packets in between.  Set by pivlfn_forward's profiling hook.
static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
void warp_corr_time_next(hipEvent_t start, hipEvent_t stop) { g_ev_start = start; g_ev_stop = stop; }

struct WcParams {
    const float *f1, *f2, *flow;
    float *out;
    float scale;
    int B, C, H, W, s, Ho, Wo, leaky;
    int dbg;      // ablation mask for tools/bench_ops.py (0 in production): 1 skip dot products, 2 skip gathers, 4 skip store, 8 exit at entry
    int strips;   // 1: walk the tiles in strips of 8 tile rows, column by column (wide images; see warp_corr_v3_kernel)
    unsigned long long *stamps;   // tools build only: per-workgroup phase stamps (16 u64 per record, 2 records per workgroup); nullptr in production
};

// In-kernel phase stamps (tools build, -DPIVLFN_STAMPS): wave 0 and wave 8 of every workgroup record s_memtime at the phase
// boundaries and s_memrealtime (100 MHz, chip-wide) at entry and exit, so that a launch's timeline can be laid out across CUs.
#ifdef PIVLFN_STAMPS
#define WC_STAMP_DECL                                                                             \
    unsigned long long st_[12];                                                                   \
    int st_n_ = 0;                                                                                \
    const bool st_on_ = p.stamps != nullptr && (threadIdx.x == 0 || threadIdx.x == 512);          \
    unsigned long long st_rt0_ = 0;                                                               \
    if (st_on_) st_rt0_ = __builtin_amdgcn_s_memrealtime()
#define WC_STAMP()                                                                                \
    do {                                                                                          \
        if (st_on_ && st_n_ < 12) st_[st_n_] = __builtin_amdgcn_s_memtime();                      \
        ++st_n_;                                                                                  \
    } while (0)
#define WC_STAMP_FLUSH()                                                                          \
    do {                                                                                          \
        if (st_on_) {                                                                             \
            unsigned long long *o_ = p.stamps + ((size_t)blockIdx.x * 2 + (threadIdx.x ? 1 : 0)) * 16; \
            for (int i_ = 0; i_ < 12; ++i_) o_[i_] = i_ < st_n_ ? st_[i_] : 0ull;                 \
            o_[12] = st_rt0_;                                                                     \
            o_[13] = __builtin_amdgcn_s_memrealtime();                                            \
            o_[14] = (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)); /* HW_REG_XCC_ID */ \
            o_[15] = (unsigned long long)st_n_;                                                   \
        }                                                                                         \
    } while (0)
#else
#define WC_STAMP_DECL do { } while (0)
#define WC_STAMP() do { } while (0)
#define WC_STAMP_FLUSH() do { } while (0)
#endif


template <int CC, bool NHWC>
__global__ __launch_bounds__(256) void warp_corr_kernel(const WcParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PP = CC + 4;          // LDS pixel pitch (floats)
    constexpr int Q = CC / 4;
    float *f2w = smem;                  // [NPOS][PP]
    float *f1t = smem + NPOS * PP;      // [64][PP]

    const int tiles_x = (p.Wo + TO - 1) / TO, tiles_y = (p.Ho + TO - 1) / TO;
    const int nblk = tiles_x * tiles_y * p.B;
    int bid = xcd_remap(blockIdx.x, nblk);
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int ox0 = tx * TO, oy0 = ty * TO;
    const int tid = threadIdx.x, lane = tid & 63, grp = tid >> 6;
    const int ppx = lane & 7, ppy = lane >> 3;
    const size_t img = (size_t)p.H * p.W;

    float acc[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) acc[k] = 0.f;

    for (int c0 = 0; c0 < p.C; c0 += CC) {
        if (c0) __syncthreads();
        if (NHWC) {
            const float *f2b = p.f2 + (size_t)b * img * p.C + c0;
            for (int idx = tid; idx < NPOS * Q; idx += 256) {
                const int pos = idx / Q, q = idx - pos * Q;
                const int iy = (oy0 + pos / TP - 3) * p.s, ix = (ox0 + pos % TP - 3) * p.s;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                    if (p.flow) {
                        const float2 uv = *reinterpret_cast<const float2 *>(p.flow + ((size_t)b * img + (size_t)iy * p.W + ix) * 4);
                        const Taps t = make_taps((float)ix + uv.x * p.scale, (float)iy + uv.y * p.scale, p.H, p.W);
                        if (t.o00 >= 0) v += t.w00 * *reinterpret_cast<const f32x4 *>(f2b + (size_t)t.o00 * p.C + 4 * q);
                        if (t.o01 >= 0) v += t.w01 * *reinterpret_cast<const f32x4 *>(f2b + (size_t)t.o01 * p.C + 4 * q);
                        if (t.o10 >= 0) v += t.w10 * *reinterpret_cast<const f32x4 *>(f2b + (size_t)t.o10 * p.C + 4 * q);
                        if (t.o11 >= 0) v += t.w11 * *reinterpret_cast<const f32x4 *>(f2b + (size_t)t.o11 * p.C + 4 * q);
                    } else {
                        v = *reinterpret_cast<const f32x4 *>(f2b + ((size_t)iy * p.W + ix) * p.C + 4 * q);
                    }
                }
                *reinterpret_cast<f32x4 *>(f2w + pos * PP + 4 * q) = v;
            }
            const float *f1b = p.f1 + (size_t)b * img * p.C + c0;
            for (int idx = tid; idx < 64 * Q; idx += 256) {
                const int pp = idx / Q, q = idx - pp * Q;
                const int oy = oy0 + (pp >> 3), ox = ox0 + (pp & 7);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (oy < p.Ho && ox < p.Wo)
                    v = *reinterpret_cast<const f32x4 *>(f1b + ((size_t)(oy * p.s) * p.W + ox * p.s) * p.C + 4 * q);
                *reinterpret_cast<f32x4 *>(f1t + pp * PP + 4 * q) = v;
            }
        } else {
            const float *f2b = p.f2 + ((size_t)b * p.C + c0) * img;
            for (int idx = tid; idx < NPOS * CC; idx += 256) {
                const int c = idx / NPOS, pos = idx - c * NPOS;
                const int iy = (oy0 + pos / TP - 3) * p.s, ix = (ox0 + pos % TP - 3) * p.s;
                float v = 0.f;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && c0 + c < p.C) {
                    const float *pl = f2b + (size_t)c * img;
                    if (p.flow) {
                        const float u = p.flow[((size_t)b * 2 + 0) * img + (size_t)iy * p.W + ix];
                        const float w = p.flow[((size_t)b * 2 + 1) * img + (size_t)iy * p.W + ix];
                        const Taps t = make_taps((float)ix + u * p.scale, (float)iy + w * p.scale, p.H, p.W);
                        if (t.o00 >= 0) v += t.w00 * pl[t.o00];
                        if (t.o01 >= 0) v += t.w01 * pl[t.o01];
                        if (t.o10 >= 0) v += t.w10 * pl[t.o10];
                        if (t.o11 >= 0) v += t.w11 * pl[t.o11];
                    } else {
                        v = pl[(size_t)iy * p.W + ix];
                    }
                }
                f2w[pos * PP + c] = v;
            }
            const float *f1b = p.f1 + ((size_t)b * p.C + c0) * img;
            for (int idx = tid; idx < 64 * CC; idx += 256) {
                const int c = idx >> 6, pp = idx & 63;
                const int oy = oy0 + (pp >> 3), ox = ox0 + (pp & 7);
                float v = 0.f;
                if (oy < p.Ho && ox < p.Wo && c0 + c < p.C) v = f1b[(size_t)c * img + (size_t)(oy * p.s) * p.W + ox * p.s];
                f1t[pp * PP + c] = v;
            }
        }
        __syncthreads();

        f32x4 a[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) a[q] = *reinterpret_cast<const f32x4 *>(f1t + lane * PP + 4 * q);
#pragma unroll
        for (int k = 0; k < 13; ++k) {
            const int d = grp + 4 * k;
            if (d < 49) {
                const int dy = d / 7, dx = d - dy * 7;
                const float *src = f2w + ((ppy + dy) * TP + ppx + dx) * PP;
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int q = 0; q < Q; q += 2) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4 *>(src + 4 * q);
                    const f32x4 v1 = *reinterpret_cast<const f32x4 *>(src + 4 * q + 4);
                    s0 = fmaf(a[q][0], v0[0], s0); s0 = fmaf(a[q][1], v0[1], s0);
                    s0 = fmaf(a[q][2], v0[2], s0); s0 = fmaf(a[q][3], v0[3], s0);
                    s1 = fmaf(a[q + 1][0], v1[0], s1); s1 = fmaf(a[q + 1][1], v1[1], s1);
                    s1 = fmaf(a[q + 1][2], v1[2], s1); s1 = fmaf(a[q + 1][3], v1[3], s1);
                }
                acc[k] += s0 + s1;
            }
        }
    }

    // transpose through LDS: [64 pixels][56] with exact zeros in lanes 49..55
    __syncthreads();
    float *ost = smem;
    const float cf = (float)p.C;
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        const int d = grp + 4 * k;
        if (d < 49) {
            float v = acc[k] / cf;
            if (p.leaky) v = lrelu01(v);
            ost[lane * OUTC + d] = v;
        }
    }
    if (grp == 0) {
#pragma unroll
        for (int d = 49; d < OUTC; ++d) ost[lane * OUTC + d] = 0.f;
    }
    __syncthreads();
    if (NHWC) {
        for (int idx = tid; idx < 64 * (OUTC / 4); idx += 256) {
            const int pp = idx / (OUTC / 4), q = idx - pp * (OUTC / 4);
            const int oy = oy0 + (pp >> 3), ox = ox0 + (pp & 7);
            if (oy < p.Ho && ox < p.Wo)
                *reinterpret_cast<f32x4 *>(p.out + ((size_t)(b * p.Ho + oy) * p.Wo + ox) * OUTC + 4 * q) =
                    *reinterpret_cast<const f32x4 *>(ost + pp * OUTC + 4 * q);
        }
    } else {
        for (int idx = tid; idx < 49 * 64; idx += 256) {
            const int d = idx >> 6, pp = idx & 63;
            const int oy = oy0 + (pp >> 3), ox = ox0 + (pp & 7);
            if (oy < p.Ho && ox < p.Wo)
                p.out[((size_t)(b * 49 + d) * p.Ho + oy) * p.Wo + ox] = ost[pp * OUTC + d];
        }
    }
}


// ---- channels-last kernels (the ones pivlfn_forward launches) -----------------------------------------------------
// Common structure, built for memory-level parallelism (a tile is ~200 KB of gathers behind a dependent flow read):
//   phase A  196 threads read the flow at their position and write the 4 bilinear taps (byte offset or out-of-range sentinel,
//            weight) into an LDS tap table -- ONE dependent round trip per tile instead of one per gathered vector;
//   phase B  all threads gather (position, 16-byte channel quad) items through buffer loads, blended vectors go to LDS;
//   phase C  dot products from LDS with ds_read_b128; phase D  the 56-lane output leaves as whole 16-byte lanes.
// (An earlier generation -- whole-C vectors, 4 or 8 waves, clamped-index gathers -- was 2-3x slower and is gone.)

// ---- v3: 32-channel chunks, double-buffered LDS, gathers of chunk k+1 in flight while chunk k is consumed -------------
// 512 threads (two waves per SIMD).  All gathers are buffer loads through a per-image descriptor: one 32-bit byte
// offset per load, and a tap that falls outside the image is encoded as an offset beyond num_records, for which the
// hardware range check returns zeros -- the zero padding of grid_sample / of the correlation costs no instruction.
// Per chunk a thread owns 3 gather items (position, 16-byte quad) (+ a 4th for the first 32 threads: 196*8 = 3*512+32)
// and one f1 item.  Chunk k+1's loads are issued right after chunk k's registers have been blended into LDS, so they
// are in flight during chunk k's dot products.
using i32x4 = __attribute__((ext_vector_type(4))) int;
constexpr unsigned OOB = 0x80000000u;

// LDS image of the warped tile: position-major 128-byte vectors (32 channels), quad q of position (r, c) stored at quad
// slot q ^ g, g = 2*(r&3) + ((c>>1)&1).  With lane = output pixel, every 16-lane group of a ds_read_b128 then covers
// all 16 slots of the 256-byte bank row for any displacement (dy, dx): conflict-free without padding.
__device__ __forceinline__ int swz_pos(int pos)
{
    const int r = pos / TP, c = pos - r * TP;
    return ((r & 3) * 2 + ((c >> 1) & 1)) & 7;
}

__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t rs, unsigned off)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
}

// Bilinear blend of one 16-byte quad from its four taps: w0*x0, then three fused multiply-adds in tap order.  Written with
// explicit fmaf so that every kernel rounds the same way (left to the compiler, a*b + c*d may fuse either product).
__device__ __forceinline__ f32x4 blend_taps(float w0, float w1, float w2, float w3, f32x4 x0, f32x4 x1, f32x4 x2, f32x4 x3)
{
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaf(w3, x3[e], fmaf(w2, x2[e], fmaf(w1, x1[e], __fmul_rn(w0, x0[e]))));
    return v;
}
// The same blend for the four left-over tile positions 192..195, whose taps the first kernels spread over four lanes and add
// as a pairwise tree: (w0*x0 + w1*x1) + (w2*x2 + w3*x3), every operation rounded.
__device__ __forceinline__ f32x4 blend_taps_tree(float w0, float w1, float w2, float w3, f32x4 x0, f32x4 x1, f32x4 x2, f32x4 x3)
{
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        v[e] = __fadd_rn(__fadd_rn(__fmul_rn(w0, x0[e]), __fmul_rn(w1, x1[e])), __fadd_rn(__fmul_rn(w2, x2[e]), __fmul_rn(w3, x3[e])));
    return v;
}

// sum / (float)C of the reference kernel (src/correlation.py:98-100).  For a power-of-two C (64 and 128 here) the product with
// the exact reciprocal is the same number and one instruction instead of a division sequence; `inv` = 0 otherwise.
__device__ __forceinline__ float mean_over_c(float sum, float cf, float inv) { return inv != 0.f ? sum * inv : sum / cf; }
__device__ __forceinline__ float pow2_reciprocal(int C) { return (C & (C - 1)) == 0 ? 1.f / (float)C : 0.f; }

// R2 layouts (see WC3_DOTS_R2): with lane = (column ppx, row pair rp) the 16-lane groups of a ds_read_b128 span four row pairs,
// so the quad swizzle of the warped tile keys on (r>>1)&3 instead of r&3, and the f1 tile's on the row-pair's upper bit.
template <bool R2>
__device__ __forceinline__ int swz_sel(int pos)
{
    if (!R2) return swz_pos(pos);
    const int r = pos / TP, c = pos - r * TP;
    return (((r >> 1) & 3) * 2 + ((c >> 1) & 1)) & 7;
}
__device__ __forceinline__ int swz_f1_r2(int fpp) { return ((fpp >> 1) & 3) | (((fpp >> 5) & 1) << 2); }

template <bool HASFLOW, bool R2>
__global__ __launch_bounds__(512, 4) void warp_corr_v3_kernel(const WcParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PP = 32, Q = 8;                  // unpadded 128-byte vectors, 16-byte quads XOR-swizzled (below)
    constexpr int BUF = (NPOS + 64) * PP;          // floats per (f2w, f1) buffer
    constexpr int NT = HASFLOW ? 4 : 1;
    constexpr int REM = NPOS * Q - 3 * 512;        // 32 items left for a 4th slot
    unsigned *tapo = reinterpret_cast<unsigned *>(smem + 2 * BUF);     // [4][NPOS] byte offsets (OOB = outside)
    float *tapw = reinterpret_cast<float *>(tapo + 4 * NPOS);          // [4][NPOS]

    const int tiles_x = (p.Wo + TO - 1) / TO, tiles_y = (p.Ho + TO - 1) / TO;
    const int nblk = tiles_x * tiles_y * p.B;
    int bid = xcd_remap(blockIdx.x, nblk);
    int tx, ty, b;
    if (p.strips) {
        // Wide images: one tile row's f2 footprint ((8+6)*s rows x W px x C) no longer fits an XCD's 4 MB L2, so in row-major order
        // the 6*s halo rows shared with the next tile row are gone by the time it starts.  Walk strips of 8 tile rows column by
        // column instead: the ~64 workgroups an XCD runs at a time then cover a compact 8x8 block of tiles whose halos overlap
        // while they are still resident; only the 6*s rows between strips are fetched twice.
        const int per_img = tiles_x * tiles_y;
        b = bid / per_img;
        const int t = bid - b * per_img;
        const int strip = t / (8 * tiles_x);
        const int rows = min(8, tiles_y - strip * 8);
        const int w = t - strip * 8 * tiles_x;
        tx = w / rows;
        ty = strip * 8 + (w - tx * rows);
    } else {
        tx = bid % tiles_x;
        bid /= tiles_x;
        ty = bid % tiles_y;
        b = bid / tiles_y;
    }
    const int ox0 = tx * TO, oy0 = ty * TO;
    if (p.dbg & 8) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave id, provably uniform
    const size_t img = (size_t)p.H * p.W;
    const unsigned img_bytes = (unsigned)(img * p.C * sizeof(float));
    const unsigned pix_bytes = (unsigned)p.C * 4u;
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.f1 + (size_t)b * img * p.C), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.f2 + (size_t)b * img * p.C), 0, img_bytes, 0x00020000);

    WC_STAMP_DECL;
    WC_STAMP();
    const int a_iy = (oy0 + tid / TP - 3) * p.s, a_ix = (ox0 + tid % TP - 3) * p.s;
    const bool a_in = tid < NPOS && a_iy >= 0 && a_iy < p.H && a_ix >= 0 && a_ix < p.W;
    float2 uv = {0.f, 0.f};
    if (HASFLOW && a_in) uv = *reinterpret_cast<const float2 *>(p.flow + ((size_t)b * img + (size_t)a_iy * p.W + a_ix) * 4);
    if (tid < NPOS) {
        const int iy = a_iy, ix = a_ix;
        unsigned o0 = OOB, o1 = OOB, o2 = OOB, o3 = OOB;
        float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
        if (a_in) {
            if (HASFLOW) {
                const Taps t = make_taps((float)ix + uv.x * p.scale, (float)iy + uv.y * p.scale, p.H, p.W);
                o0 = t.o00 < 0 ? OOB : (unsigned)t.o00 * pix_bytes; o1 = t.o01 < 0 ? OOB : (unsigned)t.o01 * pix_bytes;
                o2 = t.o10 < 0 ? OOB : (unsigned)t.o10 * pix_bytes; o3 = t.o11 < 0 ? OOB : (unsigned)t.o11 * pix_bytes;
                w0 = t.w00; w1 = t.w01; w2 = t.w10; w3 = t.w11;
            } else {
                o0 = (unsigned)(iy * p.W + ix) * pix_bytes;
                w0 = 1.f;
            }
        }
        tapo[0 * NPOS + tid] = o0; tapo[1 * NPOS + tid] = o1; tapo[2 * NPOS + tid] = o2; tapo[3 * NPOS + tid] = o3;
        tapw[0 * NPOS + tid] = w0; tapw[1 * NPOS + tid] = w1; tapw[2 * NPOS + tid] = w2; tapw[3 * NPOS + tid] = w3;
    }
    // this thread's items: 3 full slots + the remainder slot + one f1 quad
    const int q8 = tid & 7;                        // quad within the 32-channel chunk (same in every slot: 512 % 8 == 0)
    const int pos0 = tid >> 3;                     // slot u covers position pos0 + 64*u
    const int fpp = tid >> 3;                      // f1 pixel of this thread
    const int foy = oy0 + (fpp >> 3), fox = ox0 + (fpp & 7);
    const unsigned f1off = (foy < p.Ho && fox < p.Wo) ? (unsigned)((foy * p.s) * p.W + fox * p.s) * pix_bytes + 16u * q8 : OOB;
    __syncthreads();
    // remainder items (positions 192..195 x 8 quads = REM items): thread t < 4*REM owns tap (t&3) of item (t>>2)
    const int rem_pos = 192 + (tid >> 5), rem_q = (tid >> 2) & 7;
    const unsigned rem_off = tid < 4 * REM ? tapo[(tid & 3) * NPOS + rem_pos] + 16u * rem_q : OOB;
    const float rem_w = tid < 4 * REM ? tapw[(tid & 3) * NPOS + rem_pos] : 0.f;

    float acc[R2 ? 14 : 7];
#pragma unroll
    for (int k = 0; k < (R2 ? 14 : 7); ++k) acc[k] = 0.f;
    const int nch = p.C >> 5;
    f32x4 xa[3][NT];                  // gathered taps of this thread's three (position, quad) items
    f32x4 fa, ra;                     // f1 quad; remainder item (one TAP of one of the 32 left-over items, threads < 128)

#define WC3_ISSUE(X, XF, XR, CB)                                                                  \
    do {                                                                                          \
        const unsigned cb_ = (unsigned)(CB)*128u + 16u * q8;                                      \
        if (p.dbg & 2) {   /* ablation: no memory traffic */                                      \
            _Pragma("unroll") for (int u = 0; u < 3; ++u)                                         \
                _Pragma("unroll") for (int k = 0; k < NT; ++k) X[u][k] = f32x4{1.f, 2.f, 3.f, 4.f}; \
            XR = f32x4{1.f, 2.f, 3.f, 4.f};                                                       \
            XF = f32x4{1.f, 1.f, 1.f, 1.f};                                                       \
        } else {                                                                                  \
            _Pragma("unroll") for (int u = 0; u < 3; ++u)                                         \
                _Pragma("unroll") for (int k = 0; k < NT; ++k)                                    \
                    X[u][k] = bload(rs2, tapo[k * NPOS + pos0 + 64 * u] + cb_);                   \
            XR = bload(rs2, rem_off + (unsigned)(CB)*128u);                                       \
            XF = bload(rs1, f1off + (unsigned)(CB)*128u);                                         \
        }                                                                                         \
    } while (0)

#define WC3_COMMIT(X, XF, XR, F2W, F1T)                                                             \
    do {                                                                                          \
        _Pragma("unroll") for (int u = 0; u < 3; ++u) {                                           \
            const int pos = pos0 + 64 * u;                                                        \
            f32x4 v;                                                                              \
            if (NT == 4) v = blend_taps(tapw[pos], tapw[NPOS + pos], tapw[2 * NPOS + pos], tapw[3 * NPOS + pos], X[u][0], X[u][NT > 1 ? 1 : 0], X[u][NT > 2 ? 2 : 0], X[u][NT > 3 ? 3 : 0]); \
            else v = tapw[pos] * X[u][0];                                                         \
            *reinterpret_cast<f32x4 *>(F2W + pos * PP + 4 * (q8 ^ swz_sel<R2>(pos))) = v;         \
        }                                                                                         \
        if (grp < 2) {      /* waves 0,1: the 32 left-over items, one tap per lane, summed over each lane quad */ \
            f32x4 v;                                                                              \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                       \
                v[e] = __fmul_rn(rem_w, XR[e]);                                                   \
                v[e] = __fadd_rn(v[e], __shfl_xor(v[e], 1));                                      \
                v[e] = __fadd_rn(v[e], __shfl_xor(v[e], 2));                                      \
            }                                                                                     \
            if ((tid & 3) == 0) *reinterpret_cast<f32x4 *>(F2W + rem_pos * PP + 4 * (rem_q ^ swz_sel<R2>(rem_pos))) = v; \
        }                                                                                         \
        *reinterpret_cast<f32x4 *>(F1T + fpp * PP + 4 * (q8 ^ (R2 ? swz_f1_r2(fpp) : ((fpp >> 1) & 7)))) = XF; \
    } while (0)

#define WC3_DOTS(F2W, F1T)                                                                        \
    do {                                                                                          \
        int lane_l_ = lane;                                                                       \
        asm volatile("" : "+v"(lane_l_));   /* opaque per chunk: no hoisting of the LDS read addresses out of the loop */ \
        const int ppx = lane_l_ & 7, ppy = lane_l_ >> 3;                                          \
        const char *fb_ = reinterpret_cast<const char *>(F2W);                                    \
        _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) {      /* two 16-channel halves: 16 f1 registers live */ \
            f32x4 a[4];                                                                           \
            _Pragma("unroll") for (int q = 0; q < 4; ++q)                                         \
                a[q] = *reinterpret_cast<const f32x4 *>(F1T + lane * PP + 4 * ((4 * hf + q) ^ ((lane >> 1) & 7))); \
            _Pragma("unroll") for (int k = 0; k < 7; ++k) {                                       \
                const int d = grp + 8 * k;               /* wave-uniform (grp is an SGPR) */      \
                if (k < 6 || grp == 0) {                 /* d < 49: only wave 0 has a 7th */      \
                    const int dy = d / 7, dx = d - dy * 7;                                        \
                    const int r = ppy + dy, cx = ppx + dx;                                        \
                    const unsigned sb = (unsigned)(r * TP + cx) * 128u | (unsigned)(((r & 3) * 2 + ((cx >> 1) & 1)) << 4); \
                    float s0 = 0.f;                                                               \
                    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                               \
                        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(fb_ + (sb ^ (unsigned)(16 * (4 * hf + q)))); \
                        s0 = fmaf(a[q][0], v0[0], s0); s0 = fmaf(a[q][1], v0[1], s0);             \
                        s0 = fmaf(a[q][2], v0[2], s0); s0 = fmaf(a[q][3], v0[3], s0);             \
                    }                                                                             \
                    acc[k] += s0;                                                                 \
                    asm volatile("" : "+v"(acc[k]));  /* materialise here: the FMA chains must not sink below all the reads */ \
                }                                                                                 \
                __builtin_amdgcn_sched_barrier(0);    /* keep each displacement's LDS reads next to their FMAs */ \
            }                                                                                     \
        }                                                                                         \
    } while (0)


// R2: a lane owns two vertically adjacent output pixels (rows 2rp, 2rp+1 of column ppx) and one 16-channel half of the chunk
// (lanes 32-63 take the other half; summed across lane^32 once per tile); wave w < 7 owns the displacement column dx = w.
// One 16-byte read of the warped tile at row 2rp+t feeds displacement dy = t of the upper pixel AND dy = t-1 of the lower:
// 8 reads serve 14 displacement products (plus 2 reads for the two f1 quads), 40 reads per chunk and lane instead of 64,
// so the dot products stop being bound by LDS read bandwidth.  The eighth wave only helps with staging.
// LDS addresses: the tile image is made of 128-byte position vectors and the swizzle only permutes the eight 16-byte quads
// inside a vector, so address(q, t) = P[t>>1] ^ (q << 4) + t * (TP * 128) with four per-lane bases P (row pair j = t>>1 has swizzle
// key ((rp + j) & 3) * 2 + ((cx >> 1) & 1)) computed once per chunk: one v_xor per two reads and the row as an immediate offset,
// instead of the ~3 vector instructions per read the generic index arithmetic compiled to (a third of the phase's VALU work).
#define WC3_DOTS_R2(F2W, F1T)                                                                     \
    do {                                                                                          \
        if (grp < 7) {                                                                            \
            int lane_l_ = lane;                                                                   \
            asm volatile("" : "+v"(lane_l_));                                                     \
            const int l5_ = lane_l_ & 31, hfs_ = lane_l_ >> 5;                                    \
            const int ppx = l5_ & 7, rp = l5_ >> 3;                                               \
            const int cx = ppx + grp;                                                             \
            const int fp0 = 16 * rp + ppx;                                                        \
            const char *fb_ = reinterpret_cast<const char *>(F2W);                                \
            const char *f1b_ = reinterpret_cast<const char *>(F1T);                               \
            const unsigned lbase_ = (unsigned)(2 * rp * TP + cx) * 128u;                          \
            const unsigned cxb_ = (unsigned)((cx >> 1) & 1), h4_ = 4u * (unsigned)hfs_;           \
            unsigned pj_[4];                                                                      \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) pj_[j] = lbase_ | ((h4_ ^ ((((unsigned)(rp + j) & 3u) * 2u + cxb_) & 7u)) << 4); \
            const unsigned pf_ = (unsigned)fp0 * 128u | ((h4_ ^ (unsigned)swz_f1_r2(fp0)) << 4);  \
            /* every row is read one step ahead of the fused multiply-adds that use it (two registers quads in rotation) */ \
            f32x4 vr_[2];                                                                         \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                       \
                const f32x4 a0 = *reinterpret_cast<const f32x4 *>(f1b_ + (pf_ ^ (unsigned)(q << 4))); \
                const f32x4 a1 = *reinterpret_cast<const f32x4 *>(f1b_ + (pf_ ^ (unsigned)(q << 4)) + 8 * 128); \
                vr_[0] = *reinterpret_cast<const f32x4 *>(fb_ + (pj_[0] ^ (unsigned)(q << 4)));   \
                _Pragma("unroll") for (int t = 0; t < 8; ++t) {                                   \
                    if (t < 7) vr_[(t + 1) & 1] = *reinterpret_cast<const f32x4 *>(fb_ + (pj_[(t + 1) >> 1] ^ (unsigned)(q << 4)) + (t + 1) * (TP * 128)); \
                    __builtin_amdgcn_sched_barrier(0);                                            \
                    const f32x4 v0 = vr_[t & 1];                                                  \
                    if (t < 7) {                                                                  \
                        acc[t] = fmaf(a0[0], v0[0], acc[t]); acc[t] = fmaf(a0[1], v0[1], acc[t]); \
                        acc[t] = fmaf(a0[2], v0[2], acc[t]); acc[t] = fmaf(a0[3], v0[3], acc[t]); \
                        asm volatile("" : "+v"(acc[t]));      /* materialise here: the chains must not sink below all the reads */ \
                    }                                                                             \
                    if (t > 0) {                                                                  \
                        acc[6 + t] = fmaf(a1[0], v0[0], acc[6 + t]); acc[6 + t] = fmaf(a1[1], v0[1], acc[6 + t]); \
                        acc[6 + t] = fmaf(a1[2], v0[2], acc[6 + t]); acc[6 + t] = fmaf(a1[3], v0[3], acc[6 + t]); \
                        asm volatile("" : "+v"(acc[6 + t]));                                      \
                    }                                                                             \
                    __builtin_amdgcn_sched_barrier(0);                                            \
                }                                                                                 \
            }                                                                                     \
        }                                                                                         \
    } while (0)

    WC3_ISSUE(xa, fa, ra, 0);
#pragma unroll 1
    for (int c = 0; c < nch; ++c) {
        float *buf = smem + (c & 1) * BUF;
        WC3_COMMIT(xa, fa, ra, buf, (buf + NPOS * PP));            // waits for chunk c's loads, blends, writes LDS buffer c&1
        if (c + 1 < nch) WC3_ISSUE(xa, fa, ra, c + 1);             // next chunk in flight during the dot products below
        __syncthreads();     // buffer c&1 complete; every wave is past the dot products on buffer (c+1)&1
        if (!(p.dbg & 1)) {
            if constexpr (R2) WC3_DOTS_R2(buf, (buf + NPOS * PP));
            else WC3_DOTS(buf, (buf + NPOS * PP));
        }
    }
#undef WC3_COMMIT
#undef WC3_DOTS
#undef WC3_DOTS_R2
#undef WC3_ISSUE

    __syncthreads();
    float *ost = smem;
    const float cf = (float)p.C, cinv = pow2_reciprocal(p.C);
    if constexpr (R2) {
#pragma unroll
        for (int i = 0; i < 14; ++i) acc[i] += __shfl_xor(acc[i], 32);      // the two 16-channel halves
        if (grp < 7 && lane < 32) {
            const int fp0 = 16 * (lane >> 3) + (lane & 7);
#pragma unroll
            for (int t = 0; t < 7; ++t) {
                float v0 = mean_over_c(acc[t], cf, cinv), v1 = mean_over_c(acc[7 + t], cf, cinv);
                if (p.leaky) { v0 = lrelu01(v0); v1 = lrelu01(v1); }
                ost[fp0 * OUTC + 7 * t + grp] = v0;
                ost[(fp0 + 8) * OUTC + 7 * t + grp] = v1;
            }
        }
        if (grp == 7) {
#pragma unroll
            for (int d = 49; d < OUTC; ++d) ost[lane * OUTC + d] = 0.f;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int d = grp + 8 * k;
            if (d < 49) {
                float v = mean_over_c(acc[k], cf, cinv);
                if (p.leaky) v = lrelu01(v);
                ost[lane * OUTC + d] = v;
            }
        }
        if (grp == 0) {
#pragma unroll
            for (int d = 49; d < OUTC; ++d) ost[lane * OUTC + d] = 0.f;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < 64 * (OUTC / 4); idx += 512) {
        const int pp = idx / (OUTC / 4), q = idx - pp * (OUTC / 4);
        const int oy = oy0 + (pp >> 3), ox = ox0 + (pp & 7);
        if (oy < p.Ho && ox < p.Wo && !(p.dbg & 4))
            *reinterpret_cast<f32x4 *>(p.out + ((size_t)(b * p.Ho + oy) * p.Wo + ox) * OUTC + 4 * q) =
                *reinterpret_cast<const f32x4 *>(ost + pp * OUTC + 4 * q);
    }
    WC_STAMP();
    WC_STAMP_FLUSH();
}

template <bool HASFLOW, bool R2>
static int launch_wc3(const WcParams &p, hipStream_t st)
{
    const size_t lds = ((size_t)2 * (NPOS + 64) * 32 + 8 * NPOS) * sizeof(float);
    PIV_REQUIRE((size_t)p.H * p.W * p.C * sizeof(float) < 0x7fffffffull, "warp_corr: one image of %dx%dx%d exceeds the 2 GiB buffer-descriptor range", p.H, p.W, p.C);
    static LdsAttr attr;
    if (int rc = ensure_dyn_lds(attr, reinterpret_cast<const void *>(warp_corr_v3_kernel<HASFLOW, R2>), (int)lds)) return rc;
    const int nblk = cdiv(p.Wo, TO) * cdiv(p.Ho, TO) * p.B;
    if (g_ev_start) {
        hipExtLaunchKernelGGL((warp_corr_v3_kernel<HASFLOW, R2>), dim3(nblk), dim3(512), lds, st, g_ev_start, g_ev_stop, 0, p);
        g_ev_start = g_ev_stop = nullptr;
    } else {
        hipLaunchKernelGGL((warp_corr_v3_kernel<HASFLOW, R2>), dim3(nblk), dim3(512), lds, st, p);
    }
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}


// ---- v4: one whole CU per tile (1024 threads = 16 waves), 64-channel vectors, every gather of the tile in flight at once --
// Built for the latency-bound case the project's roofline target names (level 3 of a 1024x1024 pair: 256 tiles = one
// per CU): two dependent memory round trips (flow -> taps, taps -> gathers), three barriers, and each phase spread over
// 16 waves.  Per 64-channel chunk a thread owns 3 (position, 16-byte quad) items = 12 tap loads, one tap of a left-over
// item (positions 192..195) and one f1 quad: 14 buffer loads in flight per thread, 14 KiB per wave, 224 KiB per CU.
// LDS image: position-major 256-byte vectors, quad q of tile position (r, c) at quad slot q ^ (4*(r&3) + (c&3)); with
// lane = output pixel every 16-lane group of a ds_read_b128 covers all 16 slots of the 256-byte bank row for any (dy, dx).
__device__ __forceinline__ int swz16(int pos)
{
    const int r = pos / TP, c = pos - r * TP;
    return ((r & 3) << 2) | (c & 3);
}

template <bool HASFLOW>
__global__ __launch_bounds__(1024) void warp_corr_v4_kernel(const WcParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PP = 64;                         // floats per position vector
    constexpr int NT = HASFLOW ? 4 : 1;
    float *f2w = smem;                             // [NPOS][64]
    float *f1t = smem + NPOS * PP;                 // [64][64]
    unsigned *tapo = reinterpret_cast<unsigned *>(f1t + 64 * PP);      // [4][NPOS] byte offsets (OOB = outside)
    float *tapw = reinterpret_cast<float *>(tapo + 4 * NPOS);          // [4][NPOS]

    const int tiles_x = (p.Wo + TO - 1) / TO, tiles_y = (p.Ho + TO - 1) / TO;
    const int nblk = tiles_x * tiles_y * p.B;
    int bid = xcd_remap(blockIdx.x, nblk);
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int ox0 = tx * TO, oy0 = ty * TO;
    if (p.dbg & 8) return;
    WC_STAMP_DECL;
    WC_STAMP();                                   // 0: entry
    const int tid = threadIdx.x, lane = tid & 63;
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave id 0..15, provably uniform
    const size_t img = (size_t)p.H * p.W;
    const unsigned img_bytes = (unsigned)(img * p.C * sizeof(float));
    const unsigned pix_bytes = (unsigned)p.C * 4u;
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.f1 + (size_t)b * img * p.C), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.f2 + (size_t)b * img * p.C), 0, img_bytes, 0x00020000);

    // f1 quad of this thread (independent of the flow: issued first)
    const int q16 = tid & 15, pq = tid >> 4;        // pq: f1 pixel / first gather position of this thread
    const int foy = oy0 + (pq >> 3), fox = ox0 + (pq & 7);
    const unsigned f1off = (foy < p.Ho && fox < p.Wo) ? (unsigned)((foy * p.s) * p.W + fox * p.s) * pix_bytes + 16u * q16 : OOB;
    f32x4 xf = bload(rs1, f1off);

    // phase A: flow at this thread's position; the L2 prefetch of the tile's f2 footprint goes out behind it
    const int a_iy = (oy0 + tid / TP - 3) * p.s, a_ix = (ox0 + tid % TP - 3) * p.s;
    const bool a_in = tid < NPOS && a_iy >= 0 && a_iy < p.H && a_ix >= 0 && a_ix < p.W;
    float2 uv = {0.f, 0.f};
    // (Measured and dropped: touching the tile's f2 footprint from waves 4..15 while the flow is in flight, as an L2 prefetch.
    // The launch got 2.5 us slower in the network and 4 us slower from cold caches: with every CU asking at once the gathers
    // are bound by bytes per CU through the miss path, not by latency, and a touched line costs what a gathered one does.)
    if (HASFLOW && a_in) uv = *reinterpret_cast<const float2 *>(p.flow + ((size_t)b * img + (size_t)a_iy * p.W + a_ix) * 4);
    if (tid < NPOS) {
        const int iy = a_iy, ix = a_ix;
        unsigned o0 = OOB, o1 = OOB, o2 = OOB, o3 = OOB;
        float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
        if (a_in) {
            if (HASFLOW) {
                const Taps t = make_taps((float)ix + uv.x * p.scale, (float)iy + uv.y * p.scale, p.H, p.W);
                o0 = t.o00 < 0 ? OOB : (unsigned)t.o00 * pix_bytes; o1 = t.o01 < 0 ? OOB : (unsigned)t.o01 * pix_bytes;
                o2 = t.o10 < 0 ? OOB : (unsigned)t.o10 * pix_bytes; o3 = t.o11 < 0 ? OOB : (unsigned)t.o11 * pix_bytes;
                w0 = t.w00; w1 = t.w01; w2 = t.w10; w3 = t.w11;
            } else {
                o0 = (unsigned)(iy * p.W + ix) * pix_bytes;
                w0 = 1.f;
            }
        }
        tapo[0 * NPOS + tid] = o0; tapo[1 * NPOS + tid] = o1; tapo[2 * NPOS + tid] = o2; tapo[3 * NPOS + tid] = o3;
        tapw[0 * NPOS + tid] = w0; tapw[1 * NPOS + tid] = w1; tapw[2 * NPOS + tid] = w2; tapw[3 * NPOS + tid] = w3;
    }
    WC_STAMP();                                   // 1: flow read, taps written (waves 0..3 only do that work)
    __syncthreads();
    WC_STAMP();                                   // 2: barrier 1 passed
    // left-over items: positions 192..195 x 16 quads = 64 items; thread t < 256 owns tap (t&3) of item (t>>2)
    const int rem_pos = 192 + (tid >> 6), rem_q = (tid >> 2) & 15;
    const unsigned rem_off = tid < 256 ? tapo[(tid & 3) * NPOS + rem_pos] + 16u * rem_q : OOB;
    const float rem_w = tid < 256 ? tapw[(tid & 3) * NPOS + rem_pos] : 0.f;

    // Two fma chains per displacement, A over channels {0-15, 32-47, ...} and B over {16-31, 48-63, ...}, added once at the end:
    // the summation order of the throughput kernel (v3: a lane half per 16-channel half of every 32-channel chunk), so the two
    // kernels return the same bits and a pair's flow cannot depend on which of them its batch size selects.
    float accA[4], accB[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) accA[k] = accB[k] = 0.f;

#pragma unroll 1
    for (int c0 = 0; c0 < p.C; c0 += 64) {
        const unsigned cb = (unsigned)c0 * 4u + 16u * q16;
        f32x4 x[3][NT], xr;
        if (p.dbg & 2) {
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int k = 0; k < NT; ++k) x[u][k] = f32x4{1.f, 2.f, 3.f, 4.f};
            xr = f32x4{1.f, 2.f, 3.f, 4.f};
        } else {
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int k = 0; k < NT; ++k) x[u][k] = bload(rs2, tapo[k * NPOS + pq + 64 * u] + cb);
            xr = bload(rs2, rem_off + (unsigned)c0 * 4u);
            if (c0) xf = bload(rs1, f1off + (unsigned)c0 * 4u);
        }
        WC_STAMP();                                   // 3: gathers issued
        if (c0) __syncthreads();          // previous chunk's dot products are done with the LDS image
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int pos = pq + 64 * u;
            f32x4 v;
            if (NT == 4) v = blend_taps(tapw[pos], tapw[NPOS + pos], tapw[2 * NPOS + pos], tapw[3 * NPOS + pos], x[u][0], x[u][NT > 1 ? 1 : 0], x[u][NT > 2 ? 2 : 0], x[u][NT > 3 ? 3 : 0]);
            else v = tapw[pos] * x[u][0];
            *reinterpret_cast<f32x4 *>(f2w + pos * PP + 4 * (q16 ^ swz16(pos))) = v;
        }
        if (grp < 4) {                    // waves 0..3: the 64 left-over items, one tap per lane, summed over each lane quad
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = __fmul_rn(rem_w, xr[e]);
                v[e] = __fadd_rn(v[e], __shfl_xor(v[e], 1));
                v[e] = __fadd_rn(v[e], __shfl_xor(v[e], 2));
            }
            if ((tid & 3) == 0) *reinterpret_cast<f32x4 *>(f2w + rem_pos * PP + 4 * (rem_q ^ swz16(rem_pos))) = v;
        }
        *reinterpret_cast<f32x4 *>(f1t + pq * PP + 4 * (q16 ^ (pq & 15))) = xf;
        WC_STAMP();                                   // 4: gathers arrived, blended, written to LDS
        __syncthreads();
        WC_STAMP();                                   // 5: barrier 2 passed

        if (!(p.dbg & 1)) {
            int lane_l = lane;
            asm volatile("" : "+v"(lane_l));      // opaque per chunk: keeps the 64 LDS read addresses from being hoisted out of the loop
            const int ppx = lane_l & 7, ppy = lane_l >> 3;
            const char *fb = reinterpret_cast<const char *>(f2w);
#pragma unroll
            for (int h = 0; h < 4; ++h) {          // four 16-channel quarters: 16 f1 registers live at a time
                f32x4 a[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    a[q] = *reinterpret_cast<const f32x4 *>(f1t + lane * PP + 4 * ((4 * h + q) ^ (lane & 15)));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int d = grp + 16 * k;        // wave-uniform
                    if (k < 3 || grp == 0) {           // d < 49: only wave 0 has a 4th displacement
                        const int dy = d / 7, dx = d - dy * 7;
                        const int r = ppy + dy, cx = ppx + dx;
                        const unsigned sb = (unsigned)(r * TP + cx) * 256u | (unsigned)((((r & 3) << 2) | (cx & 3)) << 4);
                        float s0 = (h & 1) ? accB[k] : accA[k];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(fb + (sb ^ (unsigned)(16 * (4 * h + q))));
                            s0 = fmaf(a[q][0], v0[0], s0); s0 = fmaf(a[q][1], v0[1], s0);
                            s0 = fmaf(a[q][2], v0[2], s0); s0 = fmaf(a[q][3], v0[3], s0);
                        }
                        asm volatile("" : "+v"(s0));
                        if (h & 1) accB[k] = s0; else accA[k] = s0;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }

    WC_STAMP();                                   // 6: dot products done
    __syncthreads();
    WC_STAMP();                                   // 7: barrier 3 passed
    float *ost = smem;                           // [64][56], exact zeros in lanes 49..55
    const float cf = (float)p.C, cinv = pow2_reciprocal(p.C);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int d = grp + 16 * k;
        if (k < 3 || grp == 0) {
            float v = mean_over_c(accA[k] + accB[k], cf, cinv);
            if (p.leaky) v = lrelu01(v);
            ost[lane * OUTC + d] = v;
        }
    }
    if (grp == 1) {
#pragma unroll
        for (int d = 49; d < OUTC; ++d) ost[lane * OUTC + d] = 0.f;
    }
    __syncthreads();
    WC_STAMP();                                   // 8: transposed, barrier 4 passed
    if (tid < 64 * (OUTC / 4) && !(p.dbg & 4)) {
        const int pp = tid / (OUTC / 4), q = tid - pp * (OUTC / 4);
        const int oy = oy0 + (pp >> 3), ox = ox0 + (pp & 7);
        if (oy < p.Ho && ox < p.Wo)
            *reinterpret_cast<f32x4 *>(p.out + ((size_t)(b * p.Ho + oy) * p.Wo + ox) * OUTC + 4 * q) =
                *reinterpret_cast<const f32x4 *>(ost + pp * OUTC + 4 * q);
    }
    WC_STAMP();                                   // 9: stores issued
    WC_STAMP_FLUSH();
}

template <bool HASFLOW>
static int launch_wc4(const WcParams &p, hipStream_t st)
{
    const size_t lds = ((size_t)(NPOS + 64) * 64 + 8 * NPOS) * sizeof(float);
    PIV_REQUIRE((size_t)p.H * p.W * p.C * sizeof(float) < 0x7fffffffull, "warp_corr: one image of %dx%dx%d exceeds the 2 GiB buffer-descriptor range", p.H, p.W, p.C);
    static LdsAttr attr;
    if (int rc = ensure_dyn_lds(attr, reinterpret_cast<const void *>(warp_corr_v4_kernel<HASFLOW>), (int)lds)) return rc;
    const int nblk = cdiv(p.Wo, TO) * cdiv(p.Ho, TO) * p.B;
    if (g_ev_start) {
        hipExtLaunchKernelGGL((warp_corr_v4_kernel<HASFLOW>), dim3(nblk), dim3(1024), lds, st, g_ev_start, g_ev_stop, 0, p);
        g_ev_start = g_ev_stop = nullptr;
    } else {
        hipLaunchKernelGGL((warp_corr_v4_kernel<HASFLOW>), dim3(nblk), dim3(1024), lds, st, p);
    }
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

template <int CC, bool NHWC>
static int launch_wc(const WcParams &p, hipStream_t st)
{
    const size_t lds = (size_t)(NPOS + 64) * (CC + 4) * sizeof(float);
    static LdsAttr attr;
    if (lds > 64 * 1024)
        if (int rc = ensure_dyn_lds(attr, reinterpret_cast<const void *>(warp_corr_kernel<CC, NHWC>), (int)lds)) return rc;
    const int nblk = cdiv(p.Wo, TO) * cdiv(p.Ho, TO) * p.B;
    hipLaunchKernelGGL((warp_corr_kernel<CC, NHWC>), dim3(nblk), dim3(256), lds, st, p);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

static int wc_variant() { return PIV_KNOB(0); }

int launch_warp_corr(const float *f1, const float *f2, const float *flow, float flow_scale, float *out,
                     int B, int C, int H, int W, int stride, int leaky, bool nhwc, hipStream_t st)
{
    PIV_REQUIRE(f1 && f2 && out, "warp_corr: null pointer");
    PIV_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "warp_corr: empty shape B=%d C=%d H=%d W=%d", B, C, H, W);
    PIV_REQUIRE(stride >= 1 && stride <= 4, "warp_corr: stride=%d unsupported", stride);
    WcParams p{f1, f2, flow, out, flow_scale, B, C, H, W, stride, cdiv(H, stride), cdiv(W, stride), leaky, PIV_KNOB(2), 0,
               reinterpret_cast<unsigned long long *>(((unsigned long long)(unsigned)PIV_KNOB(10) << 32) | (unsigned)PIV_KNOB(9))};
    if (nhwc) {
        PIV_REQUIRE(C % 32 == 0, "warp_corr (channels-last): C=%d must be a multiple of 32", C);
        // f2 rows under one row of tiles: beyond half an XCD's L2 the row-major walk loses the halo rows between tile rows
        p.strips = (size_t)(TO + 6) * stride * W * C * sizeof(float) > (size_t)(2 << 20) && !(PIV_KNOB(1) & 32768);
        const int variant = wc_variant();
        if (variant == 1) {                       // v1 kernel, kept for A/B measurements (PIVLFN_WC_VARIANT=1)
            if (C % 64 == 0) return launch_wc<64, true>(p, st);
            return launch_wc<32, true>(p, st);
        }
        // Shipped policy: up to two tiles per CU the launch is latency-bound -> v4 (a whole CU per tile, needs C % 64 == 0);
        // beyond that throughput-bound -> v3 (512 threads, 128 VGPRs: two workgroups per CU overlap each other's phases).
        const long tiles = (long)cdiv(p.Wo, TO) * cdiv(p.Ho, TO) * B;
        if ((variant == 0 && C % 64 == 0 && tiles <= 512) || (variant == 5 && C % 64 == 0))
            return flow ? launch_wc4<true>(p, st) : launch_wc4<false>(p, st);
        if (variant == 4) return flow ? launch_wc3<true, false>(p, st) : launch_wc3<false, false>(p, st);     // A/B: one pixel per lane
        if (variant == 0 || variant == 5 || variant == 6) return flow ? launch_wc3<true, true>(p, st) : launch_wc3<false, true>(p, st);
        PIV_REQUIRE(false, "warp_corr: unknown kernel variant %d", variant);
    }
    if (C % 64 == 0) return launch_wc<64, false>(p, st);
    return launch_wc<32, false>(p, st);   // any C: the last chunk is zero-filled past C
}

// ---- stand-alone back-warp, NCHW (src/models.py:20-35) --------------------------------------------------
__global__ __launch_bounds__(256) void backwarp_nchw_kernel(const float *__restrict__ in, const float *__restrict__ flow,
                                                            float *__restrict__ out, int B, int C, int H, int W)
{
    const size_t img = (size_t)H * W;
    const size_t total = (size_t)B * img;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int b = (int)(i / img);
        const int pix = (int)(i - (size_t)b * img);
        const int y = pix / W, x = pix - y * W;
        const float u = flow[((size_t)b * 2 + 0) * img + pix], v = flow[((size_t)b * 2 + 1) * img + pix];
        const Taps t = make_taps((float)x + u, (float)y + v, H, W);
        for (int c = 0; c < C; ++c) {
            const float *pl = in + ((size_t)b * C + c) * img;
            float r = 0.f;
            if (t.o00 >= 0) r += t.w00 * pl[t.o00];
            if (t.o01 >= 0) r += t.w01 * pl[t.o01];
            if (t.o10 >= 0) r += t.w10 * pl[t.o10];
            if (t.o11 >= 0) r += t.w11 * pl[t.o11];
            out[((size_t)b * C + c) * img + pix] = r;
        }
    }
}

int launch_backwarp_nchw(const float *in, const float *flow, float *out, int B, int C, int H, int W, hipStream_t st)
{
    PIV_REQUIRE(in && flow && out, "backwarp: null pointer");
    PIV_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "backwarp: empty shape");
    const size_t total = (size_t)B * H * W;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(backwarp_nchw_kernel, dim3(grid), dim3(256), 0, st, in, flow, out, B, C, H, W);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

}  // namespace pivlfn
