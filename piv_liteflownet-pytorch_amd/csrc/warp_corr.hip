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
#include <type_traits>
#include "common.h"

namespace pivlfn {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

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
#ifdef PIVLFN_TOOLS
    int dbg;      // ablation mask for tools/bench_ops.py (0 in production): 1 skip dot products, 2 skip gathers, 4 skip store, 8 exit at entry
#endif
    int strips;   // 1: walk the tiles in strips of 8 tile rows, column by column (wide images; see warp_corr_v3_kernel)
    int rl;       // v6: run length (vertically consecutive tiles per run), set by its launcher
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
    if (PIV_DBG(p) & 8) return;
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
        if (PIV_DBG(p) & 2) {   /* ablation: no memory traffic */                                      \
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
        if (!(PIV_DBG(p) & 1)) {
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
        if (oy < p.Ho && ox < p.Wo && !(PIV_DBG(p) & 4))
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
    if (PIV_DBG(p) & 8) return;
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
        if (PIV_DBG(p) & 2) {
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

        if (!(PIV_DBG(p) & 1)) {
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
    if (tid < 64 * (OUTC / 4) && !(PIV_DBG(p) & 4)) {
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


// ---- v6: one persistent workgroup per CU, specialised waves, sliding window over vertical runs of tiles (round 4) -----------
// Why: phase ablations of v3 showed that gathers, dot products, output store and the per-tile skeleton of two
// co-resident workgroups add up almost serially (18.6 + 11.4 + 11 + 10 us against 43 measured) although no unit of the CU is
// more than 40 % busy.  v6 gives them to different waves of ONE 1024-thread workgroup that walks the tiles of its XCD band:
//   waves 8..15  producers: per 32-channel chunk three (position, 16-byte quad) items each = 12 buffer loads in flight per lane
//                (100 KB per CU), rolling: wait for the oldest item, blend, store to LDS, issue the same item of the NEXT chunk
//                (or of the next tile: its taps are computed from a flow value fetched one phase ahead) -- the CU's load path
//                never drains, across chunk and tile boundaries alike; they also store the previous tile's output rows.
//   wave  7      helper: the f1 tile (8 loads per lane); the 4 left-over positions 192..195 are the producers' item u = 3.
//   waves 0..6   consumers: wave = displacement column dx; lane = (row half h, column j, channel quad g); a lane owns the 4
//                output rows of its half at column j and one 16-byte quad of a 16-channel plane.  One ds_read_b128 of the warped
//                tile at row r feeds the displacements dy = r - yy of all four rows: 14 reads per 112 fused multiply-adds
//                (v3: 20, v4: 40).  28 accumulators per lane, summed over the four quad lanes once per tile (DPP).
// One barrier per chunk; chunk m lives in LDS buffer m & 1; while the consumers read chunk m the producers fill chunk m + 1 and
// have chunk m + 2 in flight.  A step cannot be shorter than one memory round trip, so what a launch costs is (bytes gathered) /
// (bytes in flight per CU / latency): the second lever is to gather fewer bytes.  SLIDING WINDOW (C = 64, two chunks, so that
// chunk k of every tile lives in buffer k): a workgroup takes RUNS of vertically consecutive tiles; a tile's 14 rows of warped
// positions live in a ring of 32 rows per plane, the tile below re-uses its last 6 rows and only 8 new rows (112 of 196 positions)
// are gathered -- 0.57-0.63 x the loads, blends and LDS stores per tile.  Runs of neighbouring tile columns run on the same XCD
// at the same time (the horizontal halo is shared through its L2).
// Summation order (the one order of this file's channels-last kernels since round 4; every launch of a level uses v6, so a
// pair's bits do not depend on its batch): a value is the sum of four chains S_g, g = (c / 4) % 4, each over c = 16 k + 4 g + e
// ascending with fmaf, combined as (S_0 + S_1) + (S_2 + S_3); the four bilinear taps are blended by blend_taps (tap order).
// LDS: a chunk buffer = two 16-channel planes of warped positions [32 ring rows][14][16] + two of the f1 tile [64 pixels][16];
// each plane is padded by 64 bytes so that the two planes an 8-lane store group writes fall on different bank halves; every
// ds_read_b128 group of the consumers covers four positions whose 64-byte vectors are 64 banks apart: conflict-free.
constexpr int K6_RING = 32;                                // position rows per plane (power of two; >= 14 + 8)
constexpr int K6_ROW = TP * 64;                            // bytes of one position row in a 16-channel plane
constexpr int K6_PLANE = K6_RING * K6_ROW + 64;
constexpr int K6_F1PLANE = 64 * 64 + 64;
constexpr int K6_F1OFF = 2 * K6_PLANE;
constexpr int K6_BUFDUMP = 2 * K6_PLANE + 2 * K6_F1PLANE;  // where items without a position store (1 KB at the end of each buffer)
constexpr int K6_BUF = K6_BUFDUMP + 1024;                  // 66816
constexpr int K6_TR = 2 * K6_BUF;                          // output tile [64 pixels][56] floats
constexpr int K6_TAPS = K6_TR + 64 * OUTC * 4;             // tap tables of two tiles: [2][196 positions]{4 byte offsets, 4 weights}
constexpr int K6_TAPTBL = NPOS * 32;
constexpr int K6_LDS = K6_TAPS + 2 * K6_TAPTBL;            // 160512 (of 163840)

struct K6Sched { int tiles_x, tiles_y, rl, rpc, r0, rstep, nruns, slide; };
// Work unit of a workgroup: a tile, or (sliding kernel) the PRIMING unit in front of a run's first tile, which gathers position
// rows 0..5 of that tile (84 positions) while every tile unit gathers rows 6..13 (112 positions) -- so that all phases of the
// sliding kernel move at most 128 positions (two items per lane) and are one code shape.
struct K6Iter { int ri, i, len, rb, n, prime; };           // run (index in this workgroup's list), tile in run, run length, ring base row, units before this one
struct K6Tile { int b, oy0, ox0, prime, rb, par; };        // par: which tap table

__device__ __forceinline__ void k6_run(const K6Sched &S, int ri, int &b, int &rr, int &tx)
{
    const int id = S.r0 + ri * S.rstep;
    tx = id % S.tiles_x;
    const int t = id / S.tiles_x;
    rr = t % S.rpc;
    b = t / S.rpc;
}
__device__ __forceinline__ K6Iter k6_begin(const K6Sched &S)
{
    K6Iter it{0, 0, 1, 0, 0, S.slide};
    if (S.nruns > 0) { int b, rr, tx; k6_run(S, 0, b, rr, tx); it.len = min(S.rl, S.tiles_y - rr * S.rl); }
    return it;
}
__device__ __forceinline__ K6Tile k6_tile(const K6Sched &S, const K6Iter &it)
{
    int b, rr, tx;
    k6_run(S, it.ri, b, rr, tx);
    return K6Tile{b, (rr * S.rl + it.i) * TO, tx * TO, it.prime, it.rb, it.n & 1};
}
__device__ __forceinline__ K6Iter k6_next(const K6Sched &S, K6Iter it)      // the caller checks ri < nruns before using the result
{
    ++it.n;
    if (it.prime) { it.prime = 0; return it; }
    if (it.i + 1 < it.len) { ++it.i; it.rb = (it.rb + 8) & (K6_RING - 1); return it; }
    ++it.ri; it.i = 0; it.rb = 0; it.prime = S.slide;
    if (it.ri < S.nruns) { int b, rr, tx; k6_run(S, it.ri, b, rr, tx); it.len = min(S.rl, S.tiles_y - rr * S.rl); }
    return it;
}

__device__ __forceinline__ f32x4 bload_s(__amdgpu_buffer_rsrc_t rs, unsigned off, int soff)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, soff, 0));
}

__device__ __forceinline__ float dpp_xor1(float v)      // lane ^ 1 inside each quad (quad_perm [1,0,3,2])
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float v)      // lane ^ 2 inside each quad (quad_perm [2,3,0,1])
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}

// Taps of one warped position: byte offsets of the four neighbours' pixels (OOB = contributes zero) and bilinear weights.
template <bool HASFLOW>
__device__ __forceinline__ void k6_taps(const WcParams &p, unsigned pix_bytes, int iy, int ix, float2 uv, i32x4 &o, f32x4 &w)
{
    const bool in = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    if constexpr (HASFLOW) {
        const Taps t = make_taps((float)ix + uv.x * p.scale, (float)iy + uv.y * p.scale, p.H, p.W);
        o[0] = (int)((!in || t.o00 < 0) ? OOB : (unsigned)t.o00 * pix_bytes);
        o[1] = (int)((!in || t.o01 < 0) ? OOB : (unsigned)t.o01 * pix_bytes);
        o[2] = (int)((!in || t.o10 < 0) ? OOB : (unsigned)t.o10 * pix_bytes);
        o[3] = (int)((!in || t.o11 < 0) ? OOB : (unsigned)t.o11 * pix_bytes);
        w = f32x4{t.w00, t.w01, t.w10, t.w11};
    } else {
        o[0] = (int)(in ? (unsigned)(iy * p.W + ix) * pix_bytes : OOB);
        o[1] = o[2] = o[3] = (int)OOB;
        w = f32x4{1.f, 0.f, 0.f, 0.f};
    }
}

// the flow of image b through a descriptor of that image alone (scalar arithmetic per call: only one image's flow, 16 bytes per
// pixel, has to fit the descriptor's 2 GiB -- a descriptor over the whole batch refused B >= 128 at 1024 x 1024)
__device__ __forceinline__ float2 k6_flow(const WcParams &p, int b, int iy, int ix)
{
    const size_t img = (size_t)p.H * p.W;
    const __amdgpu_buffer_rsrc_t rsf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.flow ? p.flow + (size_t)b * img * 4 : p.f1), 0,
                                                                         p.flow ? (unsigned)(img * 16) : 0u, 0x00020000);
    const bool in = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    const unsigned off = in ? (unsigned)(iy * p.W + ix) * 16u : OOB;
    return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rsf, (int)off, 0, 0));
}

template <int NT>
__device__ __forceinline__ f32x4 k6_blend(const f32x4 w, const f32x4 (&x)[NT])
{
    if constexpr (NT == 4) return blend_taps(w[0], w[1], w[2], w[3], x[0], x[1], x[2], x[3]);
    else return w[0] * x[0];
}

// Dot products of one consumer lane over NPL 16-channel planes (v6: the two planes of a chunk; v7: the four of a 64-channel group).
// Lane = (tile half h, column j, channel quad g) of displacement column dx: 4 output rows x 7 displacement rows = 28 accumulators,
// kept as pairs for v_pk_fma_f32: one read of the warped tile at row r feeds output rows yy and yy + 1 of a row pair (dy = r - yy
// and r - yy - 1), the f1 values of the two rows sit side by side in a register pair (the f1 tile is stored interleaved that way),
// the warped value is broadcast: P0[i] = {acc[0][i + 1], acc[1][i]}, P1[i] = {acc[2][i + 1], acc[3][i]}; the four displacements
// without a partner (acc[0][0], acc[1][6], acc[2][0], acc[3][6]) stay scalar.  Every accumulator is one fmaf chain over its
// channels in ascending order.  64 instructions and 14 LDS reads per plane.  The reads roll five rows ahead of the multiplies,
// across planes (a register is refilled right after its last use), the next plane's f1 quads arrive half a plane early -- with
// one row ahead the phase was bound by LDS latency (measured: 4000 cycles for 128 instructions in the first v7).
struct K6Acc { f32x2 P0[6], P1[6]; float s0, s1, s2, s3; };
__device__ __forceinline__ void k6_acc_zero(K6Acc &A)
{
#pragma unroll
    for (int i = 0; i < 6; ++i) { A.P0[i] = f32x2{0.f, 0.f}; A.P1[i] = f32x2{0.f, 0.f}; }
    A.s0 = A.s1 = A.s2 = A.s3 = 0.f;
}
// f1 quads of plane pl: lds + f1a + pl * F1S + {0, 512, 1024, 1536} (row pair 2h halves 0, 1; row pair 2h + 1 halves 0, 1);
// warped row r of plane pl: lds + row[r] + pl * PLS (row: anything indexable -- an array of ring rows in v6, base + r * pitch in v7).
template <int NPL, int F1S, int PLS, class RowAddr>
__device__ __forceinline__ void k6_dots(const char *lds, unsigned f1a, const RowAddr &row, K6Acc &A)
{
    constexpr int D = 5;                   // rows in flight ahead of the multiplies (10 % D == 0: register r % D serves rows r, r + D)
    f32x4 R[D], F[4], Fn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) F[i] = *reinterpret_cast<const f32x4 *>(lds + f1a + i * 512);
#pragma unroll
    for (int i = 0; i < D; ++i) R[i] = *reinterpret_cast<const f32x4 *>(lds + row[i]);
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
        const f32x2 FA[4] = {f32x2{F[0][0], F[0][1]}, f32x2{F[0][2], F[0][3]}, f32x2{F[1][0], F[1][1]}, f32x2{F[1][2], F[1][3]}};
        const f32x2 FB[4] = {f32x2{F[2][0], F[2][1]}, f32x2{F[2][2], F[2][3]}, f32x2{F[3][0], F[3][1]}, f32x2{F[3][2], F[3][3]}};
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const f32x4 v = R[r % D];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x2 vv = f32x2{v[e], v[e]};
                if (r == 0) A.s0 = fmaf(FA[e][0], v[e], A.s0);
                if (r >= 1 && r <= 6) A.P0[r - 1] = __builtin_elementwise_fma(FA[e], vv, A.P0[r - 1]);
                if (r == 7) A.s1 = fmaf(FA[e][1], v[e], A.s1);
                if (r == 2) A.s2 = fmaf(FB[e][0], v[e], A.s2);
                if (r >= 3 && r <= 8) A.P1[r - 3] = __builtin_elementwise_fma(FB[e], vv, A.P1[r - 3]);
                if (r == 9) A.s3 = fmaf(FB[e][1], v[e], A.s3);
            }
            // refill this register with the row D further on (this plane's, or the next plane's); the next plane's f1 quads half-way
            if (r + D < 10) R[r % D] = *reinterpret_cast<const f32x4 *>(lds + row[r + D] + pl * PLS);
            else if (pl + 1 < NPL) R[r % D] = *reinterpret_cast<const f32x4 *>(lds + row[r + D - 10] + (pl + 1) * PLS);
            if (r == 4 && pl + 1 < NPL) {
#pragma unroll
                for (int i = 0; i < 4; ++i) Fn[i] = *reinterpret_cast<const f32x4 *>(lds + f1a + (pl + 1) * F1S + i * 512);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (pl + 1 < NPL) {
#pragma unroll
            for (int i = 0; i < 4; ++i) F[i] = Fn[i];
        }
    }
}
// The 28 sums of a lane as acc[yy * 7 + dy], then (S_0 + S_1) + (S_2 + S_3) over the four quad lanes as a reduce-scatter: after the
// two steps lane g holds the 7 displacement rows dy of output row yy = 2 (g & 1) + (g >> 1) of its half.
__device__ __forceinline__ void k6_reduce(const K6Acc &A, int lane, float (&r2)[7])
{
    float acc[28];
#pragma unroll
    for (int dy = 0; dy < 7; ++dy) {
        acc[dy] = dy == 0 ? A.s0 : A.P0[dy > 0 ? dy - 1 : 0][0];
        acc[7 + dy] = dy == 6 ? A.s1 : A.P0[dy < 6 ? dy : 0][1];
        acc[14 + dy] = dy == 0 ? A.s2 : A.P1[dy > 0 ? dy - 1 : 0][0];
        acc[21 + dy] = dy == 6 ? A.s3 : A.P1[dy < 6 ? dy : 0][1];
    }
    const bool b0 = lane & 1, b1 = lane & 2;
    float r1[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) {
        const float mine = b0 ? acc[14 + i] : acc[i], send = b0 ? acc[i] : acc[14 + i];
        r1[i] = mine + dpp_xor1(send);
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const float mine = b1 ? r1[7 + i] : r1[i], send = b1 ? r1[i] : r1[7 + i];
        r2[i] = mine + dpp_xor2(send);
    }
}

// Tools build: lane 0 of one wave per role (0 consumer wave 0, 1 helper, 2 producer wave 8, 3 producer wave 15) stamps s_memtime
// at three points of every step (after the barrier, after its first wait, at the end of its work): 96 stamps per role and workgroup;
// enabled by dbg bit 32 (the stamp buffer is shared with v7's records: tools/wc_probe.py).
#ifdef PIVLFN_STAMPS
#define K6_STAMP(role, idx)                                                                        \
    do {                                                                                           \
        if (p.stamps && (PIV_DBG(p) & 32) && lane == 0 && (unsigned)(idx) < 96u) p.stamps[((size_t)blockIdx.x * 4 + (role)) * 96 + (idx)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define K6_STAMP(role, idx) do { } while (0)
#endif

// NS = items per producer lane and chunk: 4 (every tile gathers all 196 positions; any C >= 64) or 2 (sliding window over runs of
// tiles; C = 64).  Same arithmetic, same bits.
template <bool HASFLOW, int NS>
__global__ __launch_bounds__(1024) void warp_corr_v6_kernel(const WcParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char *lds = reinterpret_cast<char *>(smem);
    constexpr int NT = HASFLOW ? 4 : 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // Run schedule of this workgroup.  Runs (rl vertically consecutive tiles of one tile column) are numbered with the tile
    // column fastest; workgroup (xcd = id & 7, w = id >> 3) walks runs w, w + nw, ... of the XCD's contiguous band of run ids, so the
    // workgroups running together on an XCD work on neighbouring tile columns (halos shared in its L2).  With a workgroup per run
    // the ids are remapped the same way.
    K6Sched S;
    S.tiles_x = (p.Wo + TO - 1) / TO; S.tiles_y = (p.Ho + TO - 1) / TO; S.rl = p.rl; S.slide = NS == 2;
    S.rpc = (S.tiles_y + S.rl - 1) / S.rl;
    const int nruns_all = S.tiles_x * S.rpc * p.B;
    if ((int)gridDim.x == nruns_all) {
        S.nruns = 1; S.r0 = xcd_remap(blockIdx.x, nruns_all); S.rstep = 0;
    } else {
        const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3, nw = gridDim.x >> 3;
        const int q = nruns_all >> 3, r = nruns_all & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int cnt = q + (xcd < r ? 1 : 0);
        S.nruns = w < cnt ? (cnt - w + nw - 1) / nw : 0;
        S.r0 = base + w; S.rstep = nw;
    }
    if (S.nruns == 0 || (PIV_DBG(p) & 8)) return;
    const int nch = p.C >> 5;                      // 32-channel chunks per tile (>= 2, checked by the launcher)
    const size_t img = (size_t)p.H * p.W;
    const unsigned img_bytes = (unsigned)(img * p.C * sizeof(float));
    const unsigned pix_bytes = (unsigned)p.C * 4u;

    if (wave < 7) {
        // ---------------------------------------------------------------- consumers: dot products
        const int dx = wave;
        const int g = lane & 3, j = (lane >> 2) & 7, h = lane >> 5;
        const unsigned a0 = (unsigned)((j + dx) * 64 + g * 16);
        const float cf = (float)p.C, cinv = pow2_reciprocal(p.C);
        K6Acc A;
        k6_acc_zero(A);
        const unsigned fp0 = (unsigned)(K6_F1OFF + (4 * h) * 512 + (j * 4 + g) * 16);      // row pair 2h: + half * 512; row pair 2h + 1: + 1024
        int m = 0;
#pragma unroll 1
        for (K6Iter it = k6_begin(S); it.ri < S.nruns; it = k6_next(S, it)) {
            unsigned rowoff[10];          // ring rows of this lane's 10 position rows
#pragma unroll
            for (int r = 0; r < 10; ++r) rowoff[r] = (unsigned)((it.rb + 4 * h + r) & (K6_RING - 1)) * K6_ROW + a0;
            const bool work = !it.prime;      // a priming unit only fills rows of the ring
#pragma unroll 1
            for (int k = 0; k < nch; ++k, ++m) {
                if (wave == 0) K6_STAMP(0, 3 * m - 1);
                __syncthreads();
                if (wave == 0) K6_STAMP(0, 3 * m);
                const unsigned bo = (m & 1) ? (unsigned)K6_BUF : 0u;
                if (work && !(PIV_DBG(p) & 1)) {
                    unsigned row[10];
#pragma unroll
                    for (int r = 0; r < 10; ++r) row[r] = rowoff[r] + bo;
                    k6_dots<2, K6_F1PLANE, K6_PLANE>(lds, fp0 + bo, row, A);
                }
                if (wave == 0) K6_STAMP(0, 3 * m + 1);
            }
            if (!work) continue;
            float r2[7];
            k6_reduce(A, lane, r2);
            const int yy = 2 * (lane & 1) + ((lane >> 1) & 1);
            float *tr = reinterpret_cast<float *>(lds + K6_TR) + ((4 * h + yy) * 8 + j) * OUTC + dx;
#pragma unroll
            for (int dy = 0; dy < 7; ++dy) {
                float v = mean_over_c(r2[dy], cf, cinv);
                if (p.leaky) v = lrelu01(v);
                tr[7 * dy] = v;
            }
            k6_acc_zero(A);
        }
        __syncthreads();
        return;
    }

    // descriptors: flow, features and output per image (made where the image is known)
    K6Iter cur = k6_begin(S);
    K6Tile T = k6_tile(S, cur);

    if (wave == 7) {
        // ---------------------------------------------------------------- helper: the f1 tile
        const int fq8 = lane & 7, fcol = lane >> 3;                 // f1 item e: pixel (row e, column fcol), quad fq8
        const unsigned f1dst = (unsigned)(K6_F1OFF + (fq8 >> 2) * K6_F1PLANE + (fcol * 4 + (fq8 & 3)) * 16);    // + (row pair * 2 + half) * 512: [4][2][8 columns][4 quads] x 16 bytes
        {   // exact zeros in the 7 padding lanes of the output tile, once
            float *tr = reinterpret_cast<float *>(lds + K6_TR) + lane * OUTC;
#pragma unroll
            for (int d = 49; d < OUTC; ++d) tr[d] = 0.f;
        }
        f32x4 fx[8];
        unsigned f1o;
        auto set_tile = [&](const K6Tile &t) {
            const int ox = t.ox0 + fcol;
            f1o = ox < p.Wo ? (unsigned)((t.oy0 * p.s) * p.W + ox * p.s) * pix_bytes + 16u * fq8 : OOB;
        };
        auto issue = [&](const K6Tile &t, __amdgpu_buffer_rsrc_t r1, int soff, bool valid) {
            const unsigned rowb = (unsigned)(p.s * p.W) * pix_bytes;
#pragma unroll
            for (int e = 0; e < 8; ++e) fx[e] = bload_s(r1, (valid && f1o != OOB && t.oy0 + e < p.Ho) ? f1o + e * rowb : OOB, soff);
        };
        __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.f1 + (size_t)T.b * img * p.C), 0, img_bytes, 0x00020000);
        set_tile(T);
        issue(T, rs1, 0, true);
        int m = 0;
#pragma unroll 1
        while (cur.ri < S.nruns) {
            const K6Iter nxt = k6_next(S, cur);
            const bool has_next = nxt.ri < S.nruns;
            const K6Tile Tn = has_next ? k6_tile(S, nxt) : T;
            const __amdgpu_buffer_rsrc_t rs1n = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.f1 + (size_t)Tn.b * img * p.C), 0, img_bytes, 0x00020000);
#pragma unroll 1
            for (int k = 0; k < nch; ++k, ++m) {
                K6_STAMP(1, 3 * m - 1);
                if (m > 0) __syncthreads();
                K6_STAMP(1, 3 * m);
                const bool last = k == nch - 1;
                char *buf = lds + ((m & 1) ? K6_BUF : 0);
#pragma unroll
                for (int rp = 0; rp < 4; ++rp) {        // rows 2 rp, 2 rp + 1 side by side (the consumers' register pairs)
                    const f32x4 ra = fx[2 * rp], rb = fx[2 * rp + 1];
                    *reinterpret_cast<f32x4 *>(buf + f1dst + (2 * rp) * 512) = f32x4{ra[0], rb[0], ra[1], rb[1]};
                    *reinterpret_cast<f32x4 *>(buf + f1dst + (2 * rp + 1) * 512) = f32x4{ra[2], rb[2], ra[3], rb[3]};
                }
                K6_STAMP(1, 3 * m + 1);
                if (last) {
                    set_tile(Tn);
                    issue(Tn, rs1n, 0, has_next);
                } else {
                    issue(T, rs1, (k + 1) * 128, true);
                }
            }
            T = Tn; rs1 = rs1n; cur = nxt;
        }
        __syncthreads();
        __syncthreads();
        return;
    }

    // -------------------------------------------------------------------- producers: taps, gather, blend, LDS; output rows
    // Per 32-channel chunk a lane owns up to four (position, 16-byte quad) items = 16 buffer loads in flight: item u is position
    // base + 64 u + (lane id >> 3), quad (lane id & 7); a fresh tile has 196 positions (4 items, the last one 4 positions wide), a
    // tile below its predecessor 112 (2 items).  Taps: the 32 positions a WAVE gathers are computed once, by its lanes 0..31, from
    // a flow value fetched at the start of the tile's first phase, and go through a table in LDS (32 bytes per position; only this
    // wave reads its entries); per item and phase a lane reads its weights and offsets from there.  (First version: every lane
    // computed the taps of its three items itself -- ~4000 cycles of index arithmetic per tile on the critical path, phase stamps.)
    const int tp = tid - 512, wp = wave - 8;
    f32x4 x[NS][NT];
    unsigned dst[NS], tof[NS];                       // per item: LDS destination (buffer-relative) and tap-table entry of this unit
    float2 fl = {0.f, 0.f};
    auto launder = [](int v) { asm volatile("" : "+v"(v)); return v; };
    // Output rows of tile t from the LDS image: 896 quads over 512 lanes, as buffer stores whose offset is out of range for lanes
    // (or phases: `on`) with nothing to store -- no branch, so the counted waits of the phase stay counted.
    auto store_tile = [&](int tpl, const K6Tile &t, bool on) {
        const float *tr = reinterpret_cast<const float *>(lds + K6_TR);
        // the output of the tile's image alone (scalar arithmetic per tile): one image's 56 lanes have to fit 2 GiB, not the batch's
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)t.b * p.Ho * p.Wo * OUTC, 0,
                                                                             (unsigned)((size_t)p.Ho * p.Wo * OUTC * 4), 0x00020000);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int item = min(tpl + 512 * u, 64 * (OUTC / 4) - 1);
            const int px = item / (OUTC / 4), q = item - px * (OUTC / 4);
            const int oy = t.oy0 + (px >> 3), ox = t.ox0 + (px & 7);
            const bool ok = on & (tpl + 512 * u < 64 * (OUTC / 4)) & (oy < p.Ho) & (ox < p.Wo) & !(PIV_DBG(p) & 4);      // & not &&: no short-circuit branches
            const unsigned off = ok ? (unsigned)((oy * p.Wo + ox) * OUTC + 4 * q) * 4u : OOB;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(tr + px * OUTC + 4 * q);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rso, (int)off, 0, 0);
        }
    };
    // positions of unit t: [base, lim) -- all 196 (NS = 4), or rows 0..5 (priming) / rows 6..13 (tile) of the sliding kernel
    auto pos_base = [&](const K6Tile &t) { return NS == 4 ? 0 : (t.prime ? 0 : 6 * TP); };
    auto pos_lim = [&](const K6Tile &t) { return NS == 4 ? NPOS : (t.prime ? 6 * TP : NPOS); };
    auto slot_params = [&](int tpl, const K6Tile &t, unsigned (&d)[NS], unsigned (&tf)[NS]) {
        const int q8 = tpl & 7;
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int pos = pos_base(t) + 64 * u + (tpl >> 3), pp = min(pos, NPOS - 1);
            const int r = pp / TP, c = pp - r * TP;
            d[u] = pos < pos_lim(t) ? (unsigned)((q8 >> 2) * K6_PLANE + (q8 & 3) * 16 + ((t.rb + r) & (K6_RING - 1)) * K6_ROW + c * 64)
                                    : (unsigned)(K6_BUFDUMP + (tpl & 63) * 16);
            tf[u] = (unsigned)(K6_TAPS + t.par * K6_TAPTBL + pp * 32);
        }
    };
    // the position whose taps this lane computes for unit t (8 NS lanes of the wave; the others repeat them), or >= lim
    auto tap_pos = [&](const K6Tile &t) { return pos_base(t) + 64 * ((lane & (8 * NS - 1)) >> 3) + 8 * wp + (lane & 7); };
    auto tap_flow = [&](const K6Tile &t) {
        const int pos = min(tap_pos(t), NPOS - 1);
        return k6_flow(p, t.b, (t.oy0 + pos / TP - 3) * p.s, (t.ox0 + pos % TP - 3) * p.s);
    };
    auto tap_write = [&](const K6Tile &t, float2 uv) {
        const int pos = tap_pos(t), pp = min(pos, NPOS - 1);
        i32x4 o; f32x4 w;
        k6_taps<HASFLOW>(p, pix_bytes, (t.oy0 + pp / TP - 3) * p.s, (t.ox0 + pp % TP - 3) * p.s, uv, o, w);
        if (pos < pos_lim(t)) {
            char *e = lds + K6_TAPS + t.par * K6_TAPTBL + pp * 32;
            *reinterpret_cast<i32x4 *>(e) = o;
            *reinterpret_cast<f32x4 *>(e + 16) = w;
        }
    };
    auto issue_item = [&](int tpl, int u, const i32x4 o, unsigned d, __amdgpu_buffer_rsrc_t rs, int soff, bool valid) {
        const bool ok = valid & (d < (unsigned)K6_BUFDUMP);
        const unsigned qoff = 16u * (tpl & 7);
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) x[u][kk] = bload_s(rs, ok ? (unsigned)o[kk] + qoff : OOB, soff);
    };
    __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.f2 + (size_t)T.b * img * p.C), 0, img_bytes, 0x00020000);
    // prologue: this wave's taps of the first tile, then its chunk 0 in flight
    if (HASFLOW) fl = tap_flow(T);
    tap_write(T, fl);
    slot_params(tp, T, dst, tof);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the table entries this wave wrote are read back below
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        issue_item(tp, u, *reinterpret_cast<const i32x4 *>(lds + tof[u]), dst[u], rs2, 0, !(PIV_DBG(p) & 2));
        __builtin_amdgcn_sched_barrier(0);         // item order = the phases' order: the wait counts at the loop head merge both paths
    }
    // A tile's chunks are walked as code instances (first / middle / last chunk  x  items of this tile  x  items of the next tile):
    // straight-line, with a sched_barrier between items, so that every wait is a counted vmcnt in rolling order -- wait for the
    // oldest item, blend, store to LDS, issue the same item of the next chunk (in the last chunk: of the next tile).  The table
    // entries of item u + 1 are read before item u is processed.  The first chunk's phase ends with the next tile's taps.
    int m = 0;
    K6Tile Tprev = T, Tn = T;
    __amdgpu_buffer_rsrc_t rs2n = rs2;
    bool has_next = false;
    auto phase = [&](auto first_c, auto last_c, int k, bool store_prev) {
        constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
        if (wave == 8) K6_STAMP(2, 3 * m - 1); else if (wave == 15) K6_STAMP(3, 3 * m - 1);
        if (m > 0) __syncthreads();
        if (wave == 8) K6_STAMP(2, 3 * m); else if (wave == 15) K6_STAMP(3, 3 * m);
        const int tpl = launder(tp);
        store_tile(tpl, Tprev, store_prev);         // the previous tile's last chunk was step m - 2
        if constexpr (FIRST && HASFLOW) fl = tap_flow(Tn);
        const unsigned bufbase = (m & 1) ? (unsigned)K6_BUF : 0u;
        const int soff = LAST ? 0 : (k + 1) * 128;
        const bool valid = (!LAST || has_next) && !(PIV_DBG(p) & 2);      // dbg & 2 (tools): every gather out of range -- issued, no data moved
        unsigned dstn[NS], tofn[NS];                // the next unit's item parameters (last phase: its loads go out here)
        if constexpr (LAST) slot_params(tpl, Tn, dstn, tofn);
        const unsigned (&tfi)[NS] = LAST ? tofn : tof;
        const unsigned (&di)[NS] = LAST ? dstn : dst;
        f32x4 w = *reinterpret_cast<const f32x4 *>(lds + tof[0] + 16);
        i32x4 o = *reinterpret_cast<const i32x4 *>(lds + tfi[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            f32x4 wn = w; i32x4 on = o;
            if (u + 1 < NS) {
                wn = *reinterpret_cast<const f32x4 *>(lds + tof[u + 1] + 16);
                on = *reinterpret_cast<const i32x4 *>(lds + tfi[u + 1]);
            }
            *reinterpret_cast<f32x4 *>(lds + bufbase + dst[u]) = k6_blend<NT>(w, x[u]);
            if (u == 0) { if (wave == 8) K6_STAMP(2, 3 * m + 1); else if (wave == 15) K6_STAMP(3, 3 * m + 1); }
            issue_item(tpl, u, o, di[u], LAST ? rs2n : rs2, soff, valid);
            w = wn; o = on;
            __builtin_amdgcn_sched_barrier(0);      // rolling order: left alone, the scheduler consumes all items first and issues the loads as one burst (the queue drains every phase)
        }
        if constexpr (FIRST) tap_write(Tn, fl);     // the next unit's taps (its flow went out at the top of this phase)
        if constexpr (LAST) {
#pragma unroll
            for (int u = 0; u < NS; ++u) { dst[u] = dstn[u]; tof[u] = tofn[u]; }
        }
        ++m;
    };
    using std::true_type;
    using std::false_type;
    bool prev_tile = false;                         // the unit before this one was a tile (its output rows are waiting in LDS)
#pragma unroll 1
    while (cur.ri < S.nruns) {
        const K6Iter nxt = k6_next(S, cur);
        has_next = nxt.ri < S.nruns;
        Tn = has_next ? k6_tile(S, nxt) : T;
        rs2n = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.f2 + (size_t)Tn.b * img * p.C), 0, img_bytes, 0x00020000);
        phase(true_type{}, false_type{}, 0, false);
#pragma unroll 1
        for (int k = 1; k < nch - 1; ++k) phase(false_type{}, false_type{}, k, k == 1 && prev_tile);
        phase(false_type{}, true_type{}, nch - 1, nch == 2 && prev_tile);
        prev_tile = !T.prime;
        Tprev = T; T = Tn; rs2 = rs2n; cur = nxt;
    }
    __syncthreads();
    __syncthreads();
    store_tile(launder(tp), Tprev, true);          // the last unit of a workgroup is always a tile
}

template <bool HASFLOW, int NS>
static int launch_wc6_ns(const WcParams &p, int nblk, hipStream_t st)
{
    static LdsAttr attr;
    if (int rc = ensure_dyn_lds(attr, reinterpret_cast<const void *>(warp_corr_v6_kernel<HASFLOW, NS>), K6_LDS)) return rc;
    if (g_ev_start) {
        hipExtLaunchKernelGGL((warp_corr_v6_kernel<HASFLOW, NS>), dim3(nblk), dim3(1024), K6_LDS, st, g_ev_start, g_ev_stop, 0, p);
        g_ev_start = g_ev_stop = nullptr;
    } else {
        hipLaunchKernelGGL((warp_corr_v6_kernel<HASFLOW, NS>), dim3(nblk), dim3(1024), K6_LDS, st, p);
    }
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

template <bool HASFLOW>
static int launch_wc6(WcParams p, hipStream_t st)
{
    PIV_REQUIRE((size_t)p.H * p.W * p.C * sizeof(float) < 0x7fffffffull, "warp_corr: one image of %dx%dx%d exceeds the 2 GiB buffer-descriptor range", p.H, p.W, p.C);
    // flow and output go through per-image descriptors too: no limit on the batch (round 4 refused B >= 37 at level 1 of 1024 x 1024 pairs)
    PIV_REQUIRE((size_t)p.H * p.W * 16 < 0x7fffffffull && (size_t)p.Ho * p.Wo * OUTC * 4 < 0x7fffffffull,
                "warp_corr: the flow or the output of one %dx%d image exceeds the 2 GiB buffer-descriptor range", p.H, p.W);
    const int tiles_x = cdiv(p.Wo, TO), tiles_y = cdiv(p.Ho, TO);
    const int slots = std::max(8, device_cus() / 8 * 8);          // one 16-wave workgroup per CU; a multiple of 8 keeps the XCD bands
    // Run length: the sliding window needs chunk k of every tile in buffer k (two chunks: C = 64).  The longest power-of-two run
    // that still leaves a run for every CU and spreads evenly (a run count that is a multiple of the CU count, or at least four
    // runs per CU); a function of the launch's size only -- the bits of a value do not depend on it.
    p.rl = 1;
    if (p.C == 64 && !(PIV_KNOB(1) & 65536))
        for (int rl = 2; rl <= tiles_y; rl *= 2) {
            const long n = (long)tiles_x * cdiv(tiles_y, rl) * p.B;
            if (n < slots) break;
            if (n % slots == 0 || n >= 4L * slots) p.rl = rl;
        }
    const long nruns = (long)tiles_x * cdiv(tiles_y, p.rl) * p.B;
    const int nblk = nruns <= slots ? (int)nruns : slots;
    return p.rl > 1 ? launch_wc6_ns<HASFLOW, 2>(p, nblk, st) : launch_wc6_ns<HASFLOW, 4>(p, nblk, st);
}

// ---- v7: latency kernel (round 4): one tile per CU, every gather of the tile in flight at once -------------------------------
// For launches with at most one tile per CU (level 3 of a 1024x1024 pair: 256 tiles) there is nothing to pipeline against: the
// launch is a chain flow -> taps -> gathers -> blend -> dot products -> reduce / transpose -> store on every CU at once.  v7 is
// v4's structure with the pieces round 4 measured to matter:
//   * taps per wave: the 16 positions a wave gathers are computed by its lanes 0..15 and handed over through its own entries of
//     an LDS table -- no workgroup barrier between the flow read and the gathers;
//   * dot products as in v6 (k6_dots: columns of rows, v_pk_fma_f32 on row pairs, rolling LDS reads) on 7 waves -- 256 vector
//     instructions and 56 LDS reads per 64 channels and lane where v4 spent 4700-5000 cycles on LDS-bound reads (14 waves with two
//     rows per lane need 40 % more LDS bytes and were LDS-bound at twice the time);
//   * output transpose channel-major with an odd pitch: conflict-free stores (v4 / v6: 8-way conflicts on 7 stores per lane).
// Measured and dropped: gathers in plane order (a thread = one position and one 16-byte quad of each 16-channel plane, so that the
// dot products of a plane run under the arrival of the next): 64-byte pieces instead of whole 256-byte pixels per tap -- twice
// the lines per load instruction -- took the launch from 7.5 to 17.3 us.
// Summation order = v6's (four quad-lane chains over 16-channel planes ascending, (S_0 + S_1) + (S_2 + S_3), blend_taps for
// every position): a level's bits do not depend on which of the two kernels its launch size selects.
constexpr int K7_PLANE = NPOS * 64 + 64;                   // one 16-channel plane of the warped tile
constexpr int K7_F1OFF = 4 * K7_PLANE;
constexpr int K7_TAPS = K7_F1OFF + 4 * K6_F1PLANE;
constexpr int K7_TR = K7_TAPS + NPOS * 32;                 // output tile [49 channels][65] floats
constexpr int K7_TRP = 65;
constexpr int K7_LDS = K7_TR + 49 * K7_TRP * 4 + 60;       // 86160 -> rounded below

template <bool HASFLOW>
__global__ __launch_bounds__(1024) void warp_corr_v7_kernel(const WcParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char *lds = reinterpret_cast<char *>(smem);
    constexpr int NT = HASFLOW ? 4 : 1;
    const int tiles_x = (p.Wo + TO - 1) / TO, tiles_y = (p.Ho + TO - 1) / TO;
    int bid = xcd_remap(blockIdx.x, tiles_x * tiles_y * p.B);
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y, b = bid / tiles_y;
    const int ox0 = tx * TO, oy0 = ty * TO;
    if (PIV_DBG(p) & 8) return;
    WC_STAMP_DECL;
    WC_STAMP();                                   // 0: entry
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t img = (size_t)p.H * p.W;
    const unsigned img_bytes = (unsigned)(img * p.C * sizeof(float));
    const unsigned pix_bytes = (unsigned)p.C * 4u;
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.f1 + (size_t)b * img * p.C), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.f2 + (size_t)b * img * p.C), 0, img_bytes, 0x00020000);

    // f1: threads 0..511 own (row pair rp, column, quad): rows 2 rp and 2 rp + 1 of the tile -- independent of the flow, issued first
    const int q16 = tid & 15;
    const int frp = tid >> 7, fcol = (tid >> 4) & 7;
    unsigned f1o = OOB;
    if (tid < 512 && ox0 + fcol < p.Wo) f1o = (unsigned)(((oy0 + 2 * frp) * p.s) * p.W + (ox0 + fcol) * p.s) * pix_bytes + 16u * q16;
    const unsigned f1row = (unsigned)(p.s * p.W) * pix_bytes;
    const bool f1a = f1o != OOB && oy0 + 2 * frp < p.Ho, f1b_ok = f1o != OOB && oy0 + 2 * frp + 1 < p.Ho;
    f32x4 fa = bload_s(rs1, f1a ? f1o : OOB, 0), fb = bload_s(rs1, f1b_ok ? f1o + f1row : OOB, 0);

    // taps of the 16 positions this wave gathers (item u of a thread: position (tid >> 4) + 64 u = 4 wave + (lane >> 4) + 64 u)
    {
        const int l16 = lane & 15;
        const int pos = 64 * (l16 >> 2) + 4 * wave + (l16 & 3), pp = min(pos, NPOS - 1);
        const int iy = (oy0 + pp / TP - 3) * p.s, ix = (ox0 + pp % TP - 3) * p.s;
        float2 uv = {0.f, 0.f};
        if (HASFLOW) uv = k6_flow(p, b, iy, ix);
        i32x4 o; f32x4 w;
        k6_taps<HASFLOW>(p, pix_bytes, iy, ix, uv, o, w);
        if (pos < NPOS) {
            *reinterpret_cast<i32x4 *>(lds + K7_TAPS + pp * 32) = o;
            *reinterpret_cast<f32x4 *>(lds + K7_TAPS + pp * 32 + 16) = w;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave reads back its own entries below
    }
    WC_STAMP();                                   // 1: flow read, taps in the table
    WC_STAMP();                                   // 2: (no barrier here)
    const int pq = tid >> 4;                       // item u: position pq + 64 u; u = 3 exists for pq < 4 (wave 0)
    const bool has3 = wave == 0;

    // consumers: waves 0..6 = displacement column dx, lane = (tile half h, column j, quad g)
    const int dx = wave;
    const int g = lane & 3, j = (lane >> 2) & 7, h = lane >> 5;
    K6Acc A;
    k6_acc_zero(A);

#pragma unroll 1
    for (int c0 = 0; c0 < p.C; c0 += 64) {
        f32x4 x[4][NT];
        const int soff = c0 * 4;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u < 3 || has3) {
                const int pos = min(pq + 64 * u, NPOS - 1);
                const i32x4 o = *reinterpret_cast<const i32x4 *>(lds + K7_TAPS + pos * 32);
                const bool ok = (pq + 64 * u < NPOS) & !(PIV_DBG(p) & 2);
#pragma unroll
                for (int k = 0; k < NT; ++k) x[u][k] = bload_s(rs2, ok ? (unsigned)o[k] + 16u * q16 : OOB, soff);
            }
        }
        WC_STAMP();                                   // 3: gathers issued
        if (c0) {
            fa = bload_s(rs1, f1a ? f1o : OOB, soff);
            fb = bload_s(rs1, f1b_ok ? f1o + f1row : OOB, soff);
            __syncthreads();              // the previous 64 channels' dot products are done with the LDS image
        }
        int tl = tid;
        asm volatile("" : "+v"(tl));               // LDS addresses of the blend are rebuilt here rather than kept (spilled) across the gathers
        const int q16l = tl & 15, pql = tl >> 4;
        const unsigned wq = (unsigned)((q16l >> 2) * K7_PLANE + (q16l & 3) * 16);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if ((u < 3 || has3) && pql + 64 * u < NPOS) {      // weights from the table again: 16 registers fewer across the gathers
                const f32x4 w = *reinterpret_cast<const f32x4 *>(lds + K7_TAPS + (pql + 64 * u) * 32 + 16);
                *reinterpret_cast<f32x4 *>(lds + wq + (pql + 64 * u) * 64) = k6_blend<NT>(w, x[u]);
            }
        if (tl < 512) {
            char *fd = lds + K7_F1OFF + (q16l >> 2) * K6_F1PLANE + ((tl >> 7) * 2) * 512 + (((tl >> 4) & 7) * 4 + (q16l & 3)) * 16;
            *reinterpret_cast<f32x4 *>(fd) = f32x4{fa[0], fb[0], fa[1], fb[1]};
            *reinterpret_cast<f32x4 *>(fd + 512) = f32x4{fa[2], fb[2], fa[3], fb[3]};
        }
        WC_STAMP();                                   // 4: gathers arrived, blended, in LDS
        __syncthreads();
        WC_STAMP();                                   // 5: barrier passed
        if (wave < 7 && !(PIV_DBG(p) & 1)) {
            int ll = lane;
            asm volatile("" : "+v"(ll));           // the ten row addresses are built here, not kept across the gathers
            const int g_ = ll & 3, j_ = (ll >> 2) & 7, h_ = ll >> 5;
            struct { unsigned base; __device__ unsigned operator[](int r) const { return base + r * (TP * 64); } } row{(unsigned)((4 * h_ * TP + j_ + dx) * 64 + g_ * 16)};
            k6_dots<4, K6_F1PLANE, K7_PLANE>(lds, (unsigned)(K7_F1OFF + (4 * h_) * 512 + (j_ * 4 + g_) * 16), row, A);
        }
    }

    WC_STAMP();                                   // 6: dot products done
    // reduce over the four quad lanes, then the output tile channel-major with an odd pitch: the 64 lanes of a store are 64 pixels
    float *tr = reinterpret_cast<float *>(lds + K7_TR);
    if (wave < 7) {
        const float cf = (float)p.C, cinv = pow2_reciprocal(p.C);
        float r2[7];
        k6_reduce(A, lane, r2);
        const int px = (4 * h + 2 * (lane & 1) + ((lane >> 1) & 1)) * 8 + j;
#pragma unroll
        for (int dy = 0; dy < 7; ++dy) {
            float v = mean_over_c(r2[dy], cf, cinv);
            if (p.leaky) v = lrelu01(v);
            tr[(7 * dy + dx) * K7_TRP + px] = v;
        }
    }
    WC_STAMP();                                   // 7: reduced, transposed
    __syncthreads();
    WC_STAMP();                                   // 8: barrier passed
    if (tid < 64 * (OUTC / 4) && !(PIV_DBG(p) & 4)) {
        const int px = tid / (OUTC / 4), q = tid - px * (OUTC / 4);
        const int oy = oy0 + (px >> 3), ox = ox0 + (px & 7);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 4 * q + e < 49 ? tr[min(4 * q + e, 48) * K7_TRP + px] : 0.f;
        if (oy < p.Ho && ox < p.Wo) *reinterpret_cast<f32x4 *>(p.out + ((size_t)(b * p.Ho + oy) * p.Wo + ox) * OUTC + 4 * q) = v;
    }
    WC_STAMP();                                   // 9: stores issued
    WC_STAMP_FLUSH();
}

template <bool HASFLOW>
static int launch_wc7(const WcParams &p, hipStream_t st)
{
    PIV_REQUIRE((size_t)p.H * p.W * p.C * sizeof(float) < 0x7fffffffull, "warp_corr: one image of %dx%dx%d exceeds the 2 GiB buffer-descriptor range", p.H, p.W, p.C);
    PIV_REQUIRE((size_t)p.H * p.W * 16 < 0x7fffffffull, "warp_corr: the flow of one %dx%d image exceeds the 2 GiB buffer-descriptor range", p.H, p.W);
    static LdsAttr attr;
    const int ldsb = (K7_LDS + 255) / 256 * 256;
    if (int rc = ensure_dyn_lds(attr, reinterpret_cast<const void *>(warp_corr_v7_kernel<HASFLOW>), ldsb)) return rc;
    const int nblk = cdiv(p.Wo, TO) * cdiv(p.Ho, TO) * p.B;
    if (g_ev_start) {
        hipExtLaunchKernelGGL((warp_corr_v7_kernel<HASFLOW>), dim3(nblk), dim3(1024), ldsb, st, g_ev_start, g_ev_stop, 0, p);
        g_ev_start = g_ev_stop = nullptr;
    } else {
        hipLaunchKernelGGL((warp_corr_v7_kernel<HASFLOW>), dim3(nblk), dim3(1024), ldsb, st, p);
    }
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
    WcParams p{};
    p.f1 = f1; p.f2 = f2; p.flow = flow; p.out = out; p.scale = flow_scale;
    p.B = B; p.C = C; p.H = H; p.W = W; p.s = stride; p.Ho = cdiv(H, stride); p.Wo = cdiv(W, stride); p.leaky = leaky;
    PIV_SET_DBG(p, PIV_KNOB(2));
    p.strips = 0; p.rl = 1;
    p.stamps = reinterpret_cast<unsigned long long *>(((unsigned long long)(unsigned)PIV_KNOB(10) << 32) | (unsigned)PIV_KNOB(9));
    if (nhwc) {
        PIV_REQUIRE(C % 32 == 0, "warp_corr (channels-last): C=%d must be a multiple of 32", C);
        // f2 rows under one row of tiles: beyond half an XCD's L2 the row-major walk loses the halo rows between tile rows
        p.strips = (size_t)(TO + 6) * stride * W * C * sizeof(float) > (size_t)(2 << 20) && !(PIV_KNOB(1) & 32768);
        const int variant = wc_variant();
        if (variant == 1) {                       // v1 kernel, kept for A/B measurements (PIVLFN_WC_VARIANT=1)
            if (C % 64 == 0) return launch_wc<64, true>(p, st);
            return launch_wc<32, true>(p, st);
        }
        // Shipped policy (round 4): at most one tile per CU and whole 64-channel groups -> v7 (latency kernel); otherwise v6
        // (persistent specialised waves; needs at least two 32-channel chunks).  Both sum in one order.
        const long tiles6 = (long)cdiv(p.Wo, TO) * cdiv(p.Ho, TO) * B;
        if ((variant == 0 && C % 64 == 0 && tiles6 <= device_cus()) || variant == 8) {
            PIV_REQUIRE(C % 64 == 0, "warp_corr v7: C=%d must be a multiple of 64", C);
            return flow ? launch_wc7<true>(p, st) : launch_wc7<false>(p, st);
        }
        if ((variant == 0 || variant == 9) && C >= 64) return flow ? launch_wc6<true>(p, st) : launch_wc6<false>(p, st);
        // Rounds 1-3, kept for A/B measurements in the tools build: up to two tiles per CU v4 (a whole CU per tile, C % 64 == 0),
        // beyond that v3 (512 threads, two workgroups per CU).  Their summation order differs from v6's.
        const long tiles = (long)cdiv(p.Wo, TO) * cdiv(p.Ho, TO) * B;
        if (((variant == 0 || variant == 7) && C % 64 == 0 && tiles <= 512) || (variant == 5 && C % 64 == 0))
            return flow ? launch_wc4<true>(p, st) : launch_wc4<false>(p, st);
        if (variant == 4) return flow ? launch_wc3<true, false>(p, st) : launch_wc3<false, false>(p, st);     // A/B: one pixel per lane
        if (variant == 0 || variant == 5 || variant == 6 || variant == 7) return flow ? launch_wc3<true, true>(p, st) : launch_wc3<false, true>(p, st);
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
