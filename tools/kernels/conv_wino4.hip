// 3x3 stride-1 convolution by Winograd's minimal filtering F(4x4, 3x3) on the gfx950 fp32 matrix cores (round 4).
//
// Same layers and the same arithmetic class as conv_wino.hip (fp32 operands, fp32 products, fp32 accumulation on
// v_mfma_f32_32x32x2_f32; /root/reference/src/models.py:154-160, 197-204, 236-250), with the larger tile: 36 multiplies per 4x4
// outputs and input channel instead of 64 (F(2x2)) or 144 (direct) -- 1.78x fewer matrix instructions than F(2x2):
//     Y = A^T [ (G g G^T) . (B^T d B) ] A ,   d = 6x6 input patch, g = 3x3 filter, Y = 4x4 outputs, interpolation points 0, +-1, +-2, inf:
//     B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//     G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//     A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// U = G g G^T is formed once at load time in float64 and rounded to fp32.  The transforms multiply by 2, 4, 5, 8: their rounding
// makes a layer's error against float64 ~8e-6 of max |out| (F(2x2): 4e-7, direct: 3e-7); end to end the 512 x 512 flow of the CPU
// restatement moves by 1.0e-5 px against float64 where the direct fp32 forward moves by 9.2e-6 (tolerance 4.3e-4): measured before
// the kernel was written, asserted in tests/test_gpu_wino.py and by every end-to-end oracle test.
//
// Mapping.  One workgroup = 12 waves = 32 tiles (8 across x 4 down = 32 x 16 output pixels) x 32 output channels, all 36 frequency
// planes.  Wave w = (plane row i = w >> 1, column half h = w & 1) owns planes (i, 3h .. 3h+2):
//   * row half of the transform: (B^T d)[i][c] = c0 d[r0][c] + c1 d[r1][c] + c2 d[r2][c] + d[r3][c] with four rows and three
//     coefficients that depend on i only (wave-uniform), for the five patch columns c = h .. h+4 its three planes need;
//   * column half: three fixed linear forms of those five values -> the three B operands, straight from registers;
//   * A operands (weights, packed at load time in fragment order) straight from global memory; LDS holds only the raw 8-channel
//     patch, double-buffered, de-interleaved mod 4 in both directions (pixel pitch 3 quads, row pitch 8 quads mod 16) so that the
//     stride-4 operand reads of the 16 lanes of a ds_read_b128 group fall on 16 distinct slots;
//   * 12 MFMAs per wave and 8-channel chunk for ~46 packed vector instructions; 3 waves per SIMD hide the LDS and weight latencies.
// Epilogue: every wave folds its three planes over its column half (A^T M A, column part), the parts meet in LDS, eight waves
// finish two output rows each (row part over the six plane rows), bias / LeakyReLU, 16-byte NHWC stores.
// Summation order per output value: chunks ascending, k = {j, 4+j} inside a chunk, then the fixed order of the output transform:
// independent of the grid and of the batch.
#include <algorithm>
#include <type_traits>
#include <vector>
#include "common.h"

namespace pivlfn {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int W4_PW = 34, W4_PH = 18;           // patch: 32 + 2 columns, 16 + 2 rows
// The patch is staged 16 channels (two 8-channel K chunks) at a time: 64 bytes per pixel and load instead of 32 -- with 8-channel
// staging every chunk touched all 612 128-byte lines of the patch again, the CU's L1 cannot hold them (78 KB), and the L2 -> L1 path
// (34 B/clk wanted) was what the kernel waited for: -20 % with the patch loads removed (tools/w4_ablate.sh).
constexpr int W4_PIXQ = 5;                      // quads per staged pixel: 16 channels + 4 floats of padding (odd)
constexpr int W4_ROWQ = 184;                    // quads per patch row: 34 x 5 = 170, padded to 8 mod 16
constexpr int W4_NSLOT = W4_PH * W4_PW * 4;     // 16-byte slots of one 16-channel unit: 2448
constexpr int W4_PBUF = W4_PH * W4_ROWQ + 8;    // quads per patch buffer (+ a spare record for slots past the patch): 53 KB
constexpr unsigned W4OOB = 0x80000000u;

// position of patch row y / column x in the de-interleaved image: rows y = 0, 4, 8, .. first, then 1, 5, .., ...
__device__ __forceinline__ constexpr int w4_rpos(int y) { return ((y & 3) == 0 ? 0 : (y & 3) == 1 ? 5 : (y & 3) == 2 ? 10 : 14) + (y >> 2); }
__device__ __forceinline__ constexpr int w4_cpos(int x) { return ((x & 3) == 0 ? 0 : (x & 3) == 1 ? 9 : (x & 3) == 2 ? 18 : 26) + (x >> 2); }

__device__ __forceinline__ f32x4 w4_fma(float c, f32x4 a, f32x4 b) { return __builtin_elementwise_fma(f32x4{c, c, c, c}, a, b); }

// The three B operands of a wave from the patch image at quad offset bo: row form (c0, c1, c2) over the rows behind vrow[0..3],
// then the column forms of half HH.
// The three B operands of a wave, built column by column so that the kernel can put the pieces between the MFMAs of the previous
// chunk: issue(l) puts the four LDS reads of patch column HH + l in flight, fold(l) -- one group of MFMAs later -- forms
// t(l) = c0 d[r0] + c1 d[r1] + c2 d[r2] + d[r3] from them and folds it into the column forms.  Column order 0, 2, 1, 3, 4.
template <int HH>
struct W4Xform {
    f32x4 V[3], u, v;       // u, v: column values / partial forms carried between folds
    f32x4 e[4];             // reads in flight
    __device__ __forceinline__ void issue(int l, const f32x4 *smem4, int bo, const int (&vrow)[4])      // bo: buffer + 2 * sub-chunk
    {
        const int co = w4_cpos(HH + l) * W4_PIXQ;
#pragma unroll
        for (int k = 0; k < 4; ++k) e[k] = smem4[bo + vrow[k] + co];
    }
    // HH = 0: planes 0, 1, 2 from columns 0..4: 4 t0 - 5 t2 + t4;  (t3 + t4) - 4 (t1 + t2);  (t4 - t3) + 4 (t1 - t2)
    // HH = 1: planes 3, 4, 5 from columns 1..5 (t(l) = column 1 + l): (t3 - t1) +- 2 (t2 - t0);  4 t0 - 5 t2 + t4
    __device__ __forceinline__ void fold(int l, float c0, float c1, float c2)
    {
        const f32x4 t = w4_fma(c0, e[0], w4_fma(c1, e[1], w4_fma(c2, e[2], e[3])));
        if (l == 0) {
            u = t;                                                    // t0
        } else if (l == 2) {
            if (HH == 0) { V[0] = w4_fma(-5.f, t, 4.f * u); u = t; }  // u = t2
            else { V[2] = w4_fma(-5.f, t, 4.f * u); u = t - u; }      // u = t2 - t0
        } else if (l == 1) {
            if (HH == 0) { v = t - u; u = t + u; }                    // u = t1 + t2, v = t1 - t2
            else v = t;                                               // v = t1
        } else if (l == 3) {
            if (HH == 0) { V[1] = w4_fma(-4.f, u, t); V[2] = w4_fma(4.f, v, -t); }       // + t4 each with the last column
            else { const f32x4 a = t - v; V[0] = w4_fma(2.f, u, a); V[1] = w4_fma(-2.f, u, a); }
        } else {
            if (HH == 0) { V[0] += t; V[1] += t; V[2] += t; }
            else V[2] += t;
        }
    }
};

__global__ __launch_bounds__(768) void conv_wino4_kernel(const ConvParamsW p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4 *smem4 = reinterpret_cast<f32x4 *>(smem);
    const int NB = p.cout_pad >> 5;
    const int tiles_x = (p.W + 31) >> 5, tiles_y = (p.H + 15) >> 4;
    int t = xcd_remap(blockIdx.x, gridDim.x);   // the channel blocks of a spatial tile run back to back on one XCD: its patch is fetched once into that L2
    const int nb0 = t % NB;
    t /= NB;
    const int tx0 = t % tiles_x;
    t /= tiles_x;
    const int ty0 = t % tiles_y;
    const int b = t / tiles_y;
    const int x0 = tx0 * 32, y0 = ty0 * 16;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pi = wave >> 1, ph = wave & 1;    // plane row, column half
    const int n = lane & 31, g = lane >> 5;
    const int ty = n >> 3, tx = n & 7;

    // staging slots of a 16-channel unit: slot s covers (pixel, quad) = (idx >> 2, idx & 3), idx = tid + 768 s, s = 0..3 (slots 0, 1 =
    // half A, 2, 3 = half B of the unit); buffer loads through per-image descriptors that start at the patch's first image row: a slot
    // outside the image (the zero padding), past the patch or past the source's channels gets an out-of-range offset and reads zeros;
    // slots past the patch land in a spare LDS record
    const int row0 = max(y0 - 1, 0);
    unsigned ppix[4];
    int plds[4];
    const int q4 = (tid & 3) * 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int idx = tid + 768 * s;
        const int pix = idx >> 2;
        const int py = pix / W4_PW, px = pix - py * W4_PW;
        const int iy = y0 - 1 + py, ix = x0 - 1 + px;
        const bool in = idx < W4_NSLOT && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        ppix[s] = in ? (unsigned)((iy - row0) * p.W + ix) : W4OOB;
        plds[s] = (idx < W4_NSLOT ? w4_rpos(py) * W4_ROWQ + w4_cpos(px) * W4_PIXQ : W4_PH * W4_ROWQ) + (tid & 3);
    }

    // the wave's row form: t = c0 d[r0] + c1 d[r1] + c2 d[r2] + d[r3]
    const int r0 = pi == 0 ? 0 : 1, r1 = pi == 5 ? 3 : 2, r2 = pi == 0 ? 2 : 3, r3 = pi == 0 ? 4 : (pi == 5 ? 5 : 4);
    const float c0 = (pi == 0 || pi == 2 || pi == 5) ? 4.f : (pi == 1 ? -4.f : (pi == 3 ? -2.f : 2.f));
    const float c1 = (pi == 0 || pi == 5) ? -5.f : ((pi == 1 || pi == 2) ? -4.f : -1.f);
    const float c2 = (pi == 0 || pi == 5) ? 0.f : (pi == 1 ? 1.f : (pi == 2 ? -1.f : (pi == 3 ? 2.f : -2.f)));
    int vrow[4];
    {
        const int rr[4] = {r0, r1, r2, r3};
#pragma unroll
        for (int k = 0; k < 4; ++k) vrow[k] = (w4_rpos(rr[k]) + ty) * W4_ROWQ + tx * W4_PIXQ + g;      // patch pixel (4 ty + r, 4 tx + .), quad g
    }

    f32x16 acc[3];

    // weights of (chunk, nb0, plane (pi, 3 ph + k)): 64 lanes x 4 floats, contiguous
    const float *wbase = p.wpk + (((size_t)nb0 * 36 + pi * 6 + 3 * ph) * 64 + lane) * 4;
    const size_t wchunk = (size_t)NB * 36 * 256;

    const size_t img_px = (size_t)p.H * p.W;
    __amdgpu_buffer_rsrc_t rsv[3];
    int sclv[3], sst4v[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int ss = s < p.nseg ? s : 0;
        sclv[s] = p.seg[ss].cload;
        sst4v[s] = p.seg[ss].stride * 4;
        const size_t left = ((size_t)(p.H - row0) * p.W - 1) * p.seg[ss].stride + p.seg[ss].cload;
        rsv[s] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.seg[ss].ptr + ((size_t)b * img_px + (size_t)row0 * p.W) * p.seg[ss].stride), 0,
                                                   (unsigned)min(left * 4, (size_t)0x7fffffff), 0x00020000);
    }
    // staging state: the next half to load is half `hq & 1` of the 16-channel unit at (seg, cc0); units cover every source in steps
    // of 16 channels (the last unit of a source may hold fewer: its missing quads read zeros and meet zero weights)
#ifndef W4_ABL_CT
#define W4_ABL_CT 0
#endif
#define W4_ABL(BIT) ((W4_ABL_CT & (BIT)) != 0)
    int seg = 0, cc0 = 0, ubuf = 0;      // ubuf: buffer of the unit whose halves are being loaded
    f32x4 pr[2], wA[3];
#if W4_ABL_CT
    for (int s = 0; s < 2; ++s) pr[s] = f32x4{1.f, 2.f, 3.f, 4.f};
    for (int k = 0; k < 3; ++k) wA[k] = f32x4{1.f, 2.f, 3.f, 4.f};
#endif

// half HB (compile-time: 0 = slots 0, 1; 1 = slots 2, 3) of the unit at (seg, cc0) -> pr (two slots per thread); after half 1 the
// unit advances; past the last unit the loads are out of range.  (With the half as a run-time index the slot arrays went to scratch.)
#define W4_LOADH(HB)                                                                              \
    do {                                                                                          \
        const int scl_ = seg == 0 ? sclv[0] : (seg == 1 ? sclv[1] : sclv[2]);                     \
        const int sst4_ = seg == 0 ? sst4v[0] : (seg == 1 ? sst4v[1] : sst4v[2]);                 \
        const __amdgpu_buffer_rsrc_t rs_ = seg == 0 ? rsv[0] : (seg == 1 ? rsv[1] : rsv[2]);      \
        const bool qok_ = (seg < p.nseg) & (cc0 + q4 < scl_);                                     \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                           \
            const unsigned px_ = ppix[2 * (HB) + s];                                              \
            const unsigned off_ = (qok_ & (px_ != W4OOB)) ? px_ * (unsigned)sst4_ + (unsigned)q4 * 4u : W4OOB; \
            if (!W4_ABL(8)) pr[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_, (int)off_, cc0 * 4, 0)); \
        }                                                                                         \
        if (HB) {                                                                                 \
            ubuf ^= 1;                 /* the next unit goes to the other buffer */                \
            cc0 += 16;                                                                            \
            if (cc0 >= scl_) { ++seg; cc0 = 0; }                                                  \
        }                                                                                         \
    } while (0)
// commit half HB into buffer UB
#define W4_COMMITH(HB, UB)                                                                        \
    do {                                                                                          \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) smem4[(UB) * W4_PBUF + plds[2 * (HB) + s]] = pr[s]; \
    } while (0)
#define W4_LW1(K, CH) if (!W4_ABL(4)) wA[K] = *reinterpret_cast<const f32x4 *>(wbase + (size_t)((CH) + 1 < p.nchunk ? (CH) + 1 : (CH)) * wchunk + (K) * 256)
#define W4_MF(XC, J, K) if (!W4_ABL(1)) acc[K] = __builtin_amdgcn_mfma_f32_32x32x2f32(wA[K][J], XC.V[K][J], acc[K], 0, 0, 0)      /* k-th element pair J of plane K */
#define W4_SB() __builtin_amdgcn_sched_barrier(0)
// Step k = K chunk k = sub-chunk k & 1 of unit k >> 1 (every source is packed in whole units: an odd chunk count is padded with a
// zero chunk).  Half k + 3 of the patch stream goes out first (global -> registers); the twelve MFMAs of chunk k go plane by plane,
// and a plane's weight register receives the next chunk's fragment right behind its fourth MFMA; chunk k + 1's transform (reads of
// unit (k + 1) >> 1, complete since the end of step k - 1) sits between the MFMAs column by column -- a column's four LDS reads are
// issued in front of a group and used behind it; the half is committed (into the buffer unit (k - 1) >> 1 has left); barrier.
#define W4_STEP(XC, XN, CH, HB)                                                                   \
    do {                                                                                          \
        const int bn_ = ((((CH) + 1) >> 1) & 1) * W4_PBUF + 2 * (((CH) + 1) & 1);                  \
        const int ub_ = ubuf;          /* buffer of the half loaded now (W4_LOADH flips ubuf behind half 1) */ \
        W4_LOADH(HB);                                                                             \
        if (!W4_ABL(2)) XN.issue(0, smem4, bn_, vrow);                                            \
        W4_SB();                                                                                  \
        W4_MF(XC, 0, 0); W4_MF(XC, 1, 0); W4_MF(XC, 2, 0);                                        \
        W4_SB();                                                                                  \
        if (!W4_ABL(2)) { XN.fold(0, c0, c1, c2); XN.issue(2, smem4, bn_, vrow); }                \
        W4_SB();                                                                                  \
        W4_MF(XC, 3, 0);                                                                          \
        W4_LW1(0, CH);                                                                            \
        W4_MF(XC, 0, 1); W4_MF(XC, 1, 1);                                                         \
        W4_SB();                                                                                  \
        if (!W4_ABL(2)) { XN.fold(2, c0, c1, c2); XN.issue(1, smem4, bn_, vrow); }                \
        W4_SB();                                                                                  \
        W4_MF(XC, 2, 1); W4_MF(XC, 3, 1);                                                         \
        W4_LW1(1, CH);                                                                            \
        W4_SB();                                                                                  \
        if (!W4_ABL(2)) { XN.fold(1, c0, c1, c2); XN.issue(3, smem4, bn_, vrow); }                \
        W4_SB();                                                                                  \
        W4_MF(XC, 0, 2); W4_MF(XC, 1, 2);                                                         \
        W4_SB();                                                                                  \
        if (!W4_ABL(2)) { XN.fold(3, c0, c1, c2); XN.issue(4, smem4, bn_, vrow); }                \
        W4_SB();                                                                                  \
        W4_MF(XC, 2, 2); W4_MF(XC, 3, 2);                                                         \
        W4_LW1(2, CH);                                                                            \
        W4_SB();                                                                                  \
        if (!W4_ABL(2)) XN.fold(4, c0, c1, c2);                                                   \
        if (!W4_ABL(64)) W4_COMMITH(HB, ub_);                                                     \
        if (!W4_ABL(32)) __syncthreads();                                                         \
    } while (0)

    auto run = [&](auto hh) {
        constexpr int HH = decltype(hh)::value;
        W4Xform<HH> xa, xb;
#pragma unroll
        for (int k = 0; k < 3; ++k) wA[k] = *reinterpret_cast<const f32x4 *>(wbase + k * 256);
        // prologue: halves 0, 1, 2 (unit 0, half A of unit 1) in flight together -- one memory round trip, not three (with one
        // workgroup per CU nothing else runs under it) -- then the transform of chunk 0
        {
            f32x4 p0[2], p1[2];
            W4_LOADH(0); p0[0] = pr[0]; p0[1] = pr[1];
            W4_LOADH(1); p1[0] = pr[0]; p1[1] = pr[1];
            W4_LOADH(0);
            W4_COMMITH(0, 1);
            pr[0] = p0[0]; pr[1] = p0[1]; W4_COMMITH(0, 0);
            pr[0] = p1[0]; pr[1] = p1[1]; W4_COMMITH(1, 0);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
        {
            constexpr int order[5] = {0, 2, 1, 3, 4};
#pragma unroll
            for (int k = 0; k < 5; ++k) { xa.issue(order[k], smem4, 0, vrow); xa.fold(order[k], c0, c1, c2); }
        }
        int chunk = 0;
        for (; chunk + 1 < p.nchunk; chunk += 2) {
            W4_STEP(xa, xb, chunk, 1);              // even step k: half (k + 3) & 1 = 1 of unit (k + 3) >> 1
            W4_STEP(xb, xa, chunk + 1, 0);
        }
    };
    if (ph == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
#undef W4_MF
#undef W4_SB
#undef W4_LW1
#undef W4_LOADH
#undef W4_COMMITH
#undef W4_STEP

    // ---- output transform.  acc[k][4 rg + e] = M[(pi, 3 ph + k)][cout 32 nb0 + 8 rg + 4 g + e][tile n]
    // column part in registers: R[q] = sum_k A^T[q][3 ph + k] M[k]; row part across the plane rows through LDS:
    // Y[p][q] = sum_i A^T[p][i] (R_(i,0)[q] + R_(i,1)[q])
    // two exchange areas [wave 12][q 4][lane 64] quads = 48 KB each (the patch buffers are idle by now): channel group rg writes area
    // rg & 1, a barrier, then eight waves read it while all twelve already write group rg + 1 into the other area -- one barrier per
    // channel group (an area is rewritten two barriers after it was read)
    const int oq = wave & 3, op = wave >> 2;          // waves 0..7: output column q = oq, output rows 2 op, 2 op + 1 of every tile
    const int cb = nb0 * 32 + 4 * g;
    auto put = [&](int rg) {
        f32x4 *xch = smem4 + (rg & 1) * (12 * 4 * 64);
        f32x4 m[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const f32x16 a = acc[k];
            m[k] = rg == 0 ? f32x4{a[0], a[1], a[2], a[3]} : rg == 1 ? f32x4{a[4], a[5], a[6], a[7]}
                 : rg == 2 ? f32x4{a[8], a[9], a[10], a[11]} : f32x4{a[12], a[13], a[14], a[15]};
        }
        f32x4 R[4];
        if (ph == 0) {          // planes 0, 1, 2: A^T columns (1,0,0,0), (1,1,1,1), (1,-1,1,-1)
            const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2];
            R[0] = m[0] + s12; R[1] = d12; R[2] = s12; R[3] = d12;
        } else {                // planes 3, 4, 5: (1,2,4,8), (1,-2,4,-8), (0,0,0,1)
            const f32x4 s34 = m[0] + m[1], d34 = m[0] - m[1];
            R[0] = s34; R[1] = 2.f * d34; R[2] = 4.f * s34; R[3] = w4_fma(8.f, d34, m[2]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) xch[(wave * 4 + q) * 64 + lane] = R[q];
    };
    auto get = [&](int rg) {
        if (wave >= 8) return;
        const f32x4 *xch = smem4 + (rg & 1) * (12 * 4 * 64);
        f32x4 S[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) S[i] = xch[((2 * i) * 4 + oq) * 64 + lane] + xch[((2 * i + 1) * 4 + oq) * 64 + lane];
        const f32x4 s12 = S[1] + S[2], d12 = S[1] - S[2], s34 = S[3] + S[4], d34 = S[3] - S[4];
        f32x4 ya, yb;
        if (op == 0) {
            ya = (S[0] + s12) + s34;                    // row 0: 1 1 1 1 1 0
            yb = w4_fma(2.f, d34, d12);                 // row 1: 0 1 -1 2 -2 0
        } else {
            ya = w4_fma(4.f, s34, s12);                 // row 2: 0 1 1 4 4 0
            yb = w4_fma(8.f, d34, d12) + S[5];          // row 3: 0 1 -1 8 -8 1
        }
        const f32x4 bias4 = *reinterpret_cast<const f32x4 *>(p.bias + cb + 8 * rg);
        ya += bias4; yb += bias4;
        if (p.lrelu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { ya[e] = lrelu01(ya[e]); yb[e] = lrelu01(yb[e]); }
        }
        const int ox = x0 + 4 * tx + oq, oy = y0 + 4 * ty + 2 * op;
        if (ox < p.W && cb + 8 * rg < p.cout_store) {
            if (oy < p.H) *reinterpret_cast<f32x4 *>(p.out + ((size_t)(b * p.H + oy) * p.W + ox) * p.out_stride + cb + 8 * rg) = ya;
            if (oy + 1 < p.H) *reinterpret_cast<f32x4 *>(p.out + ((size_t)(b * p.H + oy + 1) * p.W + ox) * p.out_stride + cb + 8 * rg) = yb;
        }
    };
    put(0);
    __syncthreads();
    get(0); put(1);
    __syncthreads();
    get(1); put(2);
    __syncthreads();
    get(2); put(3);
    __syncthreads();
    get(3);
}

// OIHW [cout][cin][3][3] -> F(4x4, 3x3) Winograd-domain weights in MFMA A-fragment order:
//   [chunk][n block][plane row i 6][plane col j 6][lane 64][4]: lane = (cout & 31) + 32 * k-half, element e multiplies staged
//   channel 8 * chunk_in_source + 4 * k-half + e of the chunk's source.  U = G g G^T in float64, rounded once to fp32.
void pack_conv_w4(const float *w, int cout, int cin, const int *creal, const int *cload, const int *coff, int nseg,
                  std::vector<float> &pk, int *nchunk_out)
{
    const int cp = (cout + 31) / 32 * 32, NB = cp / 32;
    int nchunk = 0;
    for (int s = 0; s < nseg; ++s) nchunk += (cload[s] + 15) / 16 * 2;       // whole 16-channel units: an odd chunk count gets a zero chunk
    pk.assign((size_t)nchunk * NB * 36 * 256, 0.f);
    static const double G[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    int chunk = 0, run = 0;
    for (int s = 0; s < nseg; ++s) {
        const int off = coff[s] >= 0 ? coff[s] : run;
        for (int c0 = 0; c0 < (cload[s] + 15) / 16 * 16; c0 += 8, ++chunk)
            for (int h = 0; h < 2; ++h)
                for (int e = 0; e < 4; ++e) {
                    const int c = c0 + 4 * h + e;
                    if (c >= creal[s]) continue;
                    for (int o = 0; o < cout; ++o) {
                        const float *gk = w + ((size_t)o * cin + off + c) * 9;
                        double tmp[6][3];
                        for (int i = 0; i < 6; ++i)
                            for (int x = 0; x < 3; ++x) tmp[i][x] = G[i][0] * gk[0 * 3 + x] + G[i][1] * gk[1 * 3 + x] + G[i][2] * gk[2 * 3 + x];
                        for (int i = 0; i < 6; ++i)
                            for (int j = 0; j < 6; ++j) {
                                const double u = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
                                const int nbk = o >> 5, ln = (o & 31) + 32 * h;
                                pk[((((size_t)chunk * NB + nbk) * 36 + i * 6 + j) * 64 + ln) * 4 + e] = (float)u;
                            }
                    }
                }
        run += creal[s];
    }
    *nchunk_out = nchunk;
}

int launch_conv_w4(const ConvParamsW &p_in, hipStream_t st)
{
    const ConvParamsW &p = p_in;
    PIV_REQUIRE(p.nseg >= 1 && p.nseg <= 3 && p.wpk && p.bias && p.out, "conv_wino4: bad arguments");
    PIV_REQUIRE(p.cout_pad % 32 == 0 && p.cout_store <= p.cout_pad && p.cout_store % 4 == 0 && p.out_stride % 4 == 0,
                "conv_wino4: cout_pad=%d cout_store=%d out_stride=%d", p.cout_pad, p.cout_store, p.out_stride);
    PIV_REQUIRE(p.B > 0 && p.H > 0 && p.W > 0 && (long)p.B * p.H * p.W < (1L << 31), "conv_wino4: bad shape");
    for (int s = 0; s < p.nseg; ++s)      // 32-bit byte offsets inside the rows of one patch (descriptors are rebased per workgroup)
        PIV_REQUIRE((long)20 * p.W * p.seg[s].stride * 4 < (1L << 31), "conv_wino4: 20 rows of source %d exceed 2 GiB", s);
    int nchunk = 0;
    for (int s = 0; s < p.nseg; ++s) {
        PIV_REQUIRE(p.seg[s].cload % 4 == 0 && p.seg[s].stride % 4 == 0 && p.seg[s].ptr, "conv_wino4: segment %d misaligned", s);
        nchunk += (p.seg[s].cload + 15) / 16 * 2;
    }
    PIV_REQUIRE(nchunk == p.nchunk, "conv_wino4: segments hold %d chunks, weights were packed for %d", nchunk, p.nchunk);
    const size_t lds = (size_t)2 * W4_PBUF * 16;          // 106 KB: two 16-channel patch buffers; the epilogue's 48 KB exchange area aliases them
    static LdsAttr attr;
    if (int rc = ensure_dyn_lds(attr, reinterpret_cast<const void *>(conv_wino4_kernel), (int)lds)) return rc;
    // One workgroup per (spatial tile, channel block).  Persistent workgroups walking XCD bands of items were measured and were
    // slower (1369 vs 1235 us on 128->128 at 1024 x 1024: the loop-carried state spills at 168 registers).
    const long blocks = (long)p.B * cdiv(p.H, 16) * cdiv(p.W, 32) * (p.cout_pad / 32);
    PIV_REQUIRE(blocks < (1L << 31), "conv_wino4: grid too large");
    hipLaunchKernelGGL(conv_wino4_kernel, dim3((unsigned)blocks), dim3(768), lds, st, p);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

}  // namespace pivlfn
