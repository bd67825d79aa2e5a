// 3x3 stride-1 convolution by Winograd F(2x2, 3x3) with every fp32 operand split EXACTLY into three bf16 pieces and the products
// formed on the 16-bit matrix cores (v_mfma_f32_32x32x16_bf16: a K-16 step in 32 cycles where v_mfma_f32_32x32x2_f32 needs 512).
//
// Same layers as conv_wino.hip (/root/reference/src/models.py:77-101, 154-160, 197-204, 236-250) and the same algorithm
//     Y = A^T [ (G g G^T) . (B^T d B) ] A
// with U = G g G^T formed in float64 and rounded once to fp32 at load time, B^T d B and A^T M A as fp32 additions.  What differs is how
// the products U . V of two fp32 numbers are formed: an fp32 value x is the exact sum of three bf16 values,
//     x = h + m + l,   h = bf16(x),  m = bf16(x - h),  l = bf16(x - h - m)      (round to nearest even; both differences are exact in fp32
//                                                                               and the last one has at most 8 significant bits)
// 8 + 8 + 8 = all 24 significand bits, and bf16 has fp32's exponent range -- no scaling, no narrower input domain (see below).  A
// bf16 x bf16 product is exact in the fp32 the matrix core accumulates in, so u . v = sum over piece pairs; of the nine pairs the
// kernel forms TERMS:
//     6: hh, hm, mh, hl, mm, lh         -- dropped: ml + lm + ll <= 2^-23 |u v| in the worst case (|m| <= 2^-8 |x|, |l| <= 2^-16 |x|),
//                                          2^-26 typically: below the rounding an fp32 fma commits on the same product
//     8: + ml, lm                       -- dropped: ll <= 2^-32 |u v|
//     9: all                            -- the exact product of the two fp32 numbers
// Input domain: every finite fp32 value whose magnitude is below bf16's largest finite value 0x7f7f0000 = 3.3895e38 (0.996 of
// FLT_MAX; above it h rounds to infinity -- the fp32 transform's own sums overflow in the same neighbourhood).  Pieces below 2^-126
// (bf16 subnormals; they only arise from |x| < 2^-110) may be flushed by the matrix core: an absolute error below 2^-126 |u|.
//
// Mapping.  One workgroup = 4 waves = ONE per SIMD (512 registers each): an 8 x 8 block of 2x2 tiles (16 x 16 pixels) x 64 output
// channels, all 16 frequency planes.  Wave i owns plane row i (as in conv_wino.hip: its planes need two of the patch's four rows,
// no wave repeats another's transform), two 32-tile blocks x two 32-channel blocks per plane: 4 x 2 x 2 accumulators = 256
// registers (AGPRs).  Why this shape and not conv_wino.hip's two workgroups per CU: the matrix core consumes a 1 KB A fragment and a
// 1 KB B fragment per 32 cycles and SIMD -- 16 x the fp32 instruction's appetite -- and the weight fragments come from L2 (66-73 GB/s
// per CU, MI355X_MICROARCH.md): each must feed two MFMAs from registers (two tile blocks), which with two channel blocks (the
// transformed operand feeds two MFMAs as well: the transform and split are vector work) is 256 accumulators per wave.
//   * K step = 16 input channels = one MFMA; LDS holds the raw fp32 patch of the step (18 x 18 pixels x 16 channels, de-interleaved
//     like conv_wino.hip's image: pixel pitch 5 quads, row pitch 104 quads = 8 mod 16 -> the 16 lanes of every ds_read_b128 group land
//     on 16 distinct 16-byte slots), two buffers, one barrier per step;
//   * per step and wave: 32 ds_read_b128, 128 additions (row combination t = d[ra] +- d[rb], then the four plane columns), 64
//     transformed values split into pieces (v_cvt_pk_bf16_f32, shift / mask, subtract: 5.5 instructions per value), 24 weight-fragment
//     loads of 1 KB (4 planes x 2 channel blocks x 3 pieces, packed at load time in fragment order), 16 TERMS MFMAs;
//   * software pipeline: the planes of a step are multiplied in the order 0, 3, 1, 2; while plane k is multiplied the operand pieces and
//     weight fragments of plane k + 1 are produced / in flight, and the row-combined columns of the NEXT step replace the current
//     ones as they die (column 0 after plane 0, 3 after plane 3, 1 and 2 after plane 1): one set of 64 column registers;
//   * epilogue as conv_wino.hip: column half of A^T M A in registers, row half across the four waves through LDS, bias / LeakyReLU,
//     16-byte NHWC stores.
// Summation order per output value: K steps ascending; inside a step the piece pairs smallest first (fixed order); the matrix core
// adds the 16 products of an instruction in its own fixed order.  Independent of the grid and of the batch.
#include <algorithm>
#include <cmath>
#include <vector>
#include "common.h"
#include "wino_common.h"

#ifndef B3_PLANE_ORDER
#define B3_PLANE_ORDER 0   // 1: plane columns of a step multiplied in the order 0, 2, 1, 3 instead of 0, 3, 1, 2 (1 % slower: see B3_JB)
#endif
#ifndef B3_GINNER
#define B3_GINNER 0      // 1: a workgroup takes the channel groups of a spatial tile one after the other (measured 1-2 % SLOWER: profiles/r06_b3_group_order_ab.log)
#endif
namespace pivlfn {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

// LDS image of one K step's patch (18 x 18 pixels x 16 channels fp32): pixel records of 4 channel quads + 1 quad of padding (odd
// pitch), rows and columns de-interleaved (even then odd) as in conv_wino.hip, row pitch 104 quads = 8 mod 16: the 16 lanes of every
// ds_read_b128 group (tile rows {0,3} x columns 0-3 and rows {1,2} x columns 4-7, or the complement) land on 16 distinct 16-byte slots.
// Staged through registers (buffer_load_dwordx4 -> ds_write_b128, four lanes per 64-byte pixel record).  LDS-DMA was built twice:
// first for the patch alone (a DMA lands ~3000 cycles after its issue, and the in-order vmcnt makes the first wait for a LATER
// weight-fragment load wait for it as well), then for everything -- weight fragments through a 96 KB LDS ring three phases ahead, the
// patch a whole step ahead, every wait written by hand (tools/kernels/conv_wino_b3_lds_ring.patch.txt): bit-identical and at parity
// with this file (profiles/r06_b3_lds_ring_ab.log).  The K loop is not waiting for latency: one wave per SIMD issues ~770 instructions
// per step at 4-10 cycles each, and the workgroups pull 8 TB/s out of the L2s.  Built with -fno-slp-vectorize
// (build.sh): v_pk_add_f32 / v_pk_fma_f32 take 11 cycles of the wave's issue time, the scalar pair 8 (tools/micro/mfma_shadow.hip).
constexpr int BPIXQ = 5;        // 16-byte quads per staged pixel: 16 channels + 4 floats of padding
constexpr int BROWQ = 104;      // quads per patch row: 18 x 5 = 90, padded to 8 mod 16
constexpr int BPH = 18;         // patch rows = columns: 8 tiles x 2 + 2
constexpr int BPBUF = BPH * BROWQ + 8;      // quads per patch buffer (+ a spare record for slots past the patch)
constexpr int BPS = 6;          // staging slots per thread: three patch rows each
constexpr int BWAVE = 4 * 2 * 3 * 64;       // u32x4 per (step, channel group, wave): [plane col 4][channel block 2][piece 3][lane 64]
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b)
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));      // v_cvt_pk_bf16_f32: round to nearest even
}

// x[0..7] -> three packed-bf16 operand fragments with x = h + m + l exactly
__device__ __forceinline__ void split8(const float *x, u32x4 &ph, u32x4 &pm, u32x4 &pl)
{
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float x0 = x[2 * p], x1 = x[2 * p + 1];
        const unsigned h = cvt_pk_bf16(x0, x1);
        const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
        const unsigned m = cvt_pk_bf16(r0, r1);
        const float l0 = r0 - __builtin_bit_cast(float, m << 16), l1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
        ph[p] = h;
        pm[p] = m;
        pl[p] = cvt_pk_bf16(l0, l1);
    }
}

// LDS: two patch buffers (both live at a tile boundary: the next tile's first two steps) and the epilogue's exchange area, which takes
// the accumulators in two halves of 64 KB (one tile block each)
constexpr int BBUF_A = 0, BBUF_B = BPBUF, BXCH = 2 * BPBUF, BLDS_QUADS = BXCH + 4096;

template <int TERMS>
__global__ __launch_bounds__(256, 1) void conv_wino_b3_kernel(const ConvParamsW p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4 *smem4 = reinterpret_cast<f32x4 *>(smem);

    // ---- this workgroup's tiles.  Persistent workgroups (one per CU): every XCD owns a contiguous band of tiles (xcd_remap's bands)
    // and its workgroups take the band's tiles round-robin, so the workgroups of an XCD work on neighbouring tiles at any time.
    // Two orders.  The flat one (production): tile id = spatial tile x NG + channel group, so the channel groups of a spatial tile run
    // on neighbouring workgroups at the same time and share the patch's lines in the XCD's L2.  B3_GINNER=1 (with at least one spatial tile
    // per workgroup) lets ONE workgroup take the channel groups of a spatial tile one after the other, so that the second group's patch
    // loads are L2 hits: a patch load that goes to HBM costs a quarter of a K step (B3_ABL_PATCHNEAR) -- but measured 1-2 % slower on
    // every layer (profiles/r06_b3_group_order_ab.log): the neighbour's concurrent request already turns the second miss into a hit.
    const int NG = p.cout_pad >> 6;             // 64-channel groups
    const int tiles_x = (p.W + 15) >> 4, tiles_y = (p.H + 15) >> 4;
    const int sp_total = p.B * tiles_y * tiles_x;
    const bool ginner = B3_GINNER && NG > 1 && sp_total >= (int)gridDim.x;
    const int units = ginner ? sp_total : sp_total * NG;      // what the bands and the round-robin count: spatial tiles or tiles
    const int xcd = blockIdx.x & 7;
    const int tstride = ((int)gridDim.x - xcd + 7) >> 3;      // workgroups on this XCD
    const int tq = units >> 3, tr = units & 7;
    const int uend = (xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq) + tq + (xcd < tr ? 1 : 0);
    const int ufirst = uend - tq - (xcd < tr ? 1 : 0) + (int)(blockIdx.x >> 3);       // this workgroup's first unit
    if (ufirst >= uend) return;
    const int ntiles = ((uend - ufirst + tstride - 1) / tstride) * (ginner ? NG : 1);       // tiles this workgroup multiplies
    const int tile0 = ginner ? ufirst * NG : ufirst;           // id of the first one (channel group fastest)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // staging: slot s (0..5) of a thread covers patch rows 3 s .. 3 s + 2: thread -> (row r = 0..2 inside the slot, column px, quad); 216 of the
    // 256 threads carry a record.  A thread's global offset is then ONE value + s x (3 image rows) and its LDS offset one of TWO values
    // (even / odd slots: the de-interleaved row order) + a constant: 4 registers where a slot table takes 12 (which the register allocator
    // spilled, and a spill reload waits vmcnt(0) -- for the patch loads in flight and for the epilogue's stores).
    // These per-lane addresses are recomputed at every tile start from a laundered thread id (B3_LANE_ADDRS): as loop invariants they
    // would be live across the epilogue, which wants the registers -- the allocator spilled four of them per tile.
    int plds_even, plds_odd, abase[2];
    // plane row i = wave: (B^T d)[i][.] = d[ra][.] + sb * d[rb][.]
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sb = wave == 1 ? 1.f : -1.f;
#define B3_LANE_ADDRS()                                                                           \
    do {                                                                                          \
        int tl_ = tid;                                                                            \
        asm volatile("" : "+v"(tl_));                                                             \
        const int spix_ = tl_ >> 2, sr_ = spix_ >= 54 ? 3 : (spix_ >= 36 ? 2 : (spix_ >= 18 ? 1 : 0)), spx_ = spix_ - 18 * sr_; \
        const int scol_ = sr_ < 3 ? ((spx_ >> 1) + (spx_ & 1) * 9) * BPIXQ + (tl_ & 3) : 90 + (tl_ & 3);      /* idle threads: the padding at the end of a row */ \
        plds_even = (sr_ < 3 ? ((sr_ >> 1) + (sr_ & 1) * 9) * BROWQ : 0) + scol_;                 \
        plds_odd = (sr_ < 3 ? (((sr_ + 1) >> 1) + (1 - (sr_ & 1)) * 9) * BROWQ : 0) + scol_;      \
        const int n_ = tl_ & 31, g_ = (tl_ >> 5) & 1;                                             \
        _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                          \
            abase[mb] = (mb * 4 + (n_ >> 3) + (ra >> 1) + (ra & 1) * (BPH / 2)) * BROWQ + (n_ & 7) * BPIXQ + 2 * g_;      /* patch pixel (2 tyl + ra, 2 txl), quads 2 g, 2 g + 1 */ \
    } while (0)
    const int abdiff = (((rb >> 1) + (rb & 1) * (BPH / 2)) - ((ra >> 1) + (ra & 1) * (BPH / 2))) * BROWQ;             // row rb from row ra (wave-uniform)

    // (starting a tile's accumulators from the constant 0 in its first MFMAs, as conv_wino.hip does, costs a second copy of the step
    //  body whose results the register allocator keeps out of the accumulation registers: 50-80 spilled registers.  They are zeroed.)
    f32x16 acc[4][2][2];
#define B3_ZERO_ACC()                                                                             \
    do {                                                                                          \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) \
            _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[j][mb][nb][r] = 0.f; \
    } while (0)
    B3_ZERO_ACC();

    // weights: one descriptor over the packed array; voffset = lane (+ piece), soffset = (step, group, wave, plane column, channel block)
    const size_t wbytes = (size_t)p.nchunk * NG * 4 * BWAVE * 16;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wpk_b), 0, (unsigned)min(wbytes, (size_t)0xffffffffu), 0x00020000);
    const int wstep = NG * 4 * BWAVE * 16;                      // bytes per K step
    const int wlane = lane * 16;
    int woff = ((tile0 % NG) * 4 + wave) * BWAVE * 16;          // this wave's fragments of the current step

    // ---- the load stream: (tile, step) whose patch is fetched next, two steps ahead of the matrix work and across tile boundaries
    const size_t img_px = (size_t)p.H * p.W;
    int lleft = ntiles - 1, lseg = 0, lc0 = 0, ly0 = 0, lx0 = 0, lrow0 = 0;      // lleft: tiles the load stream has in front of it
    int lscl = p.seg[0].cload;      // staged channels of the load stream's source -- kept in a scalar register: as p.seg[lseg].cload it was an
                                    // s_load per step whose lgkmcnt(0) also waited for every LDS read in flight
    // tile coordinates without divisions in the loop: (channel group, tile column, tile row, image) of the load stream, advanced by the
    // decomposed stride with carries; the tile being multiplied is the one the load stream left at its last B3_LNEXT
    int lng = tile0 % NG, ltx = (tile0 / NG) % tiles_x, lty = (tile0 / NG / tiles_x) % tiles_y, lb = tile0 / NG / tiles_x / tiles_y;
    // the stride between two tiles of this workgroup, decomposed: flat order = tstride tile ids; ginner = the next channel group of the
    // same spatial tile, then tstride spatial tiles on
    const int dsp = ginner ? tstride : tstride / NG;
    const int dng = ginner ? 0 : tstride % NG, dtx = dsp % tiles_x, dty = (dsp / tiles_x) % tiles_y, db = dsp / tiles_x / tiles_y;
    int cng = lng, cx0 = ltx * 16, cy0 = lty * 16, cb = lb;
    __amdgpu_buffer_rsrc_t rsv[3];
    // byte offset of the thread's record in slot 1..5 (+ s x lrstep) and in slot 0, inside the load tile's current source (the descriptor
    // starts one row above the tile: at the patch's first row); WOOB = none (left / right / top padding, idle thread); rows below the image
    // fall behind the descriptor's end
    unsigned pvo_base = WOOB, pvo_first = WOOB;
    int lrstep = 0;
#define B3_PVO(SEG)                                                                               \
    do {                                                                                          \
        const int sst4_ = p.seg[SEG].stride * 4;                                                  \
        int tid_ = tid;                                                                           \
        asm volatile("" : "+v"(tid_));     /* laundered: nothing derived from it is kept in registers as a loop invariant */ \
        const int spix_ = tid_ >> 2, sr_ = spix_ >= 54 ? 3 : (spix_ >= 36 ? 2 : (spix_ >= 18 ? 1 : 0)), ix_ = lx0 - 1 + spix_ - 18 * sr_; \
        lrstep = 3 * p.W * sst4_;                                                                 \
        pvo_base = sr_ < 3 && ix_ >= 0 && ix_ < p.W ? (unsigned)(sr_ * p.W + ix_) * (unsigned)sst4_ + (unsigned)(tid_ & 3) * 16u : WOOB; \
        pvo_first = ly0 == 0 && sr_ == 0 ? WOOB : pvo_base;      /* patch row 0 of a tile in the first image row: the zero padding */ \
    } while (0)
// descriptors of the load tile: they start at the first image row of its patch (64-bit scalar arithmetic), the 32-bit lane offsets span
// the patch rows only
#define B3_LTILE()                                                                                \
    do {                                                                                          \
        lx0 = ltx * 16;                                                                           \
        ly0 = lty * 16;                                                                           \
        const int lb_ = lb;                                                                       \
        lrow0 = ly0 - 1;                                                                          \
        const long pix0_ = (long)lb_ * (long)img_px + (long)lrow0 * p.W;       /* first pixel of the patch's first image row */ \
        const size_t pixl_ = (size_t)(p.H - lrow0) * p.W - 1;                   /* pixels from there to the image's last one */ \
        _Pragma("unroll") for (int s = 0; s < 3; ++s) {                                           \
            if (s == 0 || s < p.nseg) {     /* single-source layers (most) pay for one descriptor, not three */ \
                const size_t left = pixl_ * p.seg[s].stride + p.seg[s].cload;                     \
                rsv[s] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.seg[s].ptr) + pix0_ * p.seg[s].stride, 0, \
                                                           (unsigned)min(left * 4, (size_t)0x7fffffff), 0x00020000); \
            } else {                                                                              \
                rsv[s] = rsv[0];                                                                  \
            }                                                                                     \
        }                                                                                         \
        lseg = 0; lc0 = 0; lscl = p.seg[0].cload;                                                 \
        B3_PVO(0);                                                                                \
    } while (0)
// patch of the load stream's step -> PR, then advance the stream inside its tile (the loads and this rare branch sit at the top of a
// step: everything behind them is one basic block).  The stream changes tiles in ONE place, B3_LNEXT, two steps before the matrix work does.
#if B3_BRANCHFREE     // the source switch as selects + an unconditional B3_PVO: the whole step is one basic block
#define B3_ADVANCE_SEG(SCL)                                                                       \
    do {                                                                                          \
        const bool sw_ = lc0 >= (SCL) && lseg + 1 < p.nseg;                                       \
        lseg += sw_ ? 1 : 0;                                                                      \
        lscl = p.seg[lseg].cload;                                                                 \
        lc0 = sw_ ? 0 : lc0;                                                                      \
        B3_PVO(lseg);                                                                             \
    } while (0)
#else
#define B3_ADVANCE_SEG(SCL)                                                                       \
    do {                                                                                          \
        if (lc0 >= (SCL) && lseg + 1 < p.nseg) {                                                  \
            ++lseg;                                                                               \
            lscl = p.seg[lseg].cload;                                                             \
            lc0 = 0;                                                                              \
            B3_PVO(lseg);                                                                         \
        }                                                                                         \
    } while (0)
#endif
#define B3_LOADP(PR)                                                                              \
    do {                                                                                          \
        const int scl_ = lscl;                                                                    \
        const __amdgpu_buffer_rsrc_t rs_ = lseg == 0 ? rsv[0] : (lseg == 1 ? rsv[1] : rsv[2]);    \
        int tq_ = tid;                                                                            \
        asm volatile("" : "+v"(tq_));       /* (tid & 3) * 4 kept as a loop invariant was the one register the allocator still spilled */ \
        const bool qok_ = lc0 + (tq_ & 3) * 4 < scl_ B3_ABL_PATCHCOND;                            \
        _Pragma("unroll") for (int s = 0; s < BPS; ++s)                                           \
            PR[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_, (int)(qok_ ? B3_ABL_NEAR(s ? pvo_base + (unsigned)(s * lrstep) : pvo_first) : WOOB), lc0 * 4, 0)); \
        lc0 += 16;                                                                                \
        B3_ADVANCE_SEG(scl_);                                                                     \
    } while (0)
#define B3_COMMIT(PR, BOFF)                                                                       \
    do {                                                                                          \
        _Pragma("unroll") for (int s = 0; s < BPS; ++s)                                           \
            smem4[(BOFF) + ((s & 1) ? plds_odd + ((3 * s - 1) / 2) * BROWQ : plds_even + (3 * s / 2) * BROWQ)] = PR[s]; \
    } while (0)
// the load stream moves on to this workgroup's next tile (past the last one: every slot out of range -- zeros, no memory access)
#define B3_LNEXT()                                                                                \
    do {                                                                                          \
        cng = lng; cx0 = lx0; cy0 = ly0; cb = lb;                                                 \
        if (lleft > 0) {                                                                          \
            --lleft;                                                                              \
            if (ginner && lng + 1 < NG) {                                                         \
                ++lng;              /* the next channel group of the same spatial tile: same patch, same descriptors */ \
            } else {                                                                              \
                lng = ginner ? 0 : lng + dng;                                                     \
                if (lng >= NG) { lng -= NG; ++ltx; }                                              \
                ltx += dtx;                                                                       \
                if (ltx >= tiles_x) { ltx -= tiles_x; ++lty; }                                    \
                lty += dty;                                                                       \
                if (lty >= tiles_y) { lty -= tiles_y; ++lb; }                                     \
                lb += db;                                                                         \
            }                                                                                     \
            B3_LTILE();                                                                           \
        } else {                                                                                  \
            lseg = 0; lc0 = 0;                                                                    \
            pvo_base = pvo_first = WOOB;                                                          \
        }                                                                                         \
    } while (0)
// patch column C (0..3) of both tile blocks, rows ra / rb combined, from the image at quad offset BOFF
#define B3_TCOL(C, BOFF)                                                                          \
    do {                                                                                          \
        _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) _Pragma("unroll") for (int q = 0; q < 2; ++q) { \
            const f32x4 a_ = smem4[(BOFF) + abase[mb] + (((C) >> 1) + ((C)&1) * 9) * BPIXQ + q];  \
            const f32x4 b_ = smem4[(BOFF) + abase[mb] + abdiff + (((C) >> 1) + ((C)&1) * 9) * BPIXQ + q]; \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) t[mb][C][q][e] = __builtin_fmaf(sb, b_[e], a_[e]); \
        }                                                                                         \
    } while (0)
// plane column J of both tile blocks from t, split into pieces -> V[SLOT]
#define B3_VPLANE(J, SLOT)                                                                        \
    do {                                                                                          \
        _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) {                                        \
            float v_[8];                                                                          \
            _Pragma("unroll") for (int q = 0; q < 2; ++q) _Pragma("unroll") for (int e = 0; e < 4; ++e) \
                v_[4 * q + e] = (J) == 0 ? t[mb][0][q][e] - t[mb][2][q][e]                        \
                              : (J) == 1 ? t[mb][1][q][e] + t[mb][2][q][e]                        \
                              : (J) == 2 ? t[mb][2][q][e] - t[mb][1][q][e] : t[mb][1][q][e] - t[mb][3][q][e]; \
            split8(v_, V[SLOT][mb][0], V[SLOT][mb][1], V[SLOT][mb][2]);                           \
        }                                                                                         \
    } while (0)
// weight fragments of plane column J of the step at byte offset WOFF -> U[SLOT]
#define B3_ULOAD(WOFF, J, SLOT)                                                                   \
    do {                                                                                          \
        _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) _Pragma("unroll") for (int pc = 0; pc < 3; ++pc) \
            U[SLOT][nb][pc] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlane + pc * 1024, B3_ABL_UOFF((WOFF) + (J)*6144) + nb * 3072, 0)); \
    } while (0)
#define B3_ONE(J, SLOT, WP, XP)                                                                   \
    _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) \
        acc[J][mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, U[SLOT][nb][WP]), \
                                                                 __builtin_bit_cast(bf16x8, V[SLOT][mb][XP]), acc[J][mb][nb], 0, 0, 0)
// (weight piece, operand piece), smallest products first
#define B3_MFMAS(J, SLOT)                                                                         \
    do {                                                                                          \
        if (TERMS >= 9) { B3_ONE(J, SLOT, 2, 2); }                                                \
        if (TERMS >= 8) { B3_ONE(J, SLOT, 2, 1); B3_ONE(J, SLOT, 1, 2); }                         \
        B3_ONE(J, SLOT, 2, 0); B3_ONE(J, SLOT, 1, 1); B3_ONE(J, SLOT, 0, 2);                      \
        B3_ONE(J, SLOT, 1, 0); B3_ONE(J, SLOT, 0, 1); B3_ONE(J, SLOT, 0, 0);                      \
    } while (0)
// One K step = four phases, one per plane in the order 0, 3, 1, 2.  A phase starts by waiting for its weight fragments (issued one
// phase ago) and only then issues the next plane's: every vector-memory wait of the loop is "everything issued so far", which is
// what the compiler makes of any wait while an LDS-DMA is in flight anyway.  Under the phase's MFMAs the next plane's operand pieces
// are produced and the next step's columns replace the current ones as they die; phase 0 also sends the patch of the step after next
// on its way to LDS.  WNEXT = byte offset of the next step's weight fragments.
#ifdef B3_ABL_NOPATCH       // timing build: every patch load out of range (zeros, no memory access)
#define B3_ABL_PATCHCOND && p.H < 0
#else
#define B3_ABL_PATCHCOND
#endif
#ifdef B3_ABL_PATCHNEAR     // timing build: every patch load inside a small window of its descriptor (cache hits, data still random)
#ifndef B3_ABL_NEARMASK
#define B3_ABL_NEARMASK 0x3ff0u      // 16 KB: L1 hits; 0x1ffff0u = 2 MB: L2 hits
#endif
#define B3_ABL_NEAR(V) ((V) >= WOOB ? WOOB : ((V) & B3_ABL_NEARMASK) + (unsigned)lrstep)      /* three rows below the descriptor base: inside the image */
#else
#define B3_ABL_NEAR(V) (V)
#endif
#ifdef B3_ABL_UNEAR         // timing build: every weight-fragment load from this wave's first 6 KB (L1 hits, data still random)
#define B3_ABL_UOFF(V) (wave * BWAVE * 16)
#else
#define B3_ABL_UOFF(V) (V)
#endif
#ifndef B3_PATCH_POS
// behind the fragment loads of phase 1 with six terms (1.4-2.6 % faster than phase 2, -3 % and more against position 0: tools/bench_b3.py,
// same box, profiles/r06_b3_patch_pos_ab.log); the eight- and nine-term instances spill there and keep phase 2 -- no spill in any instance
#define B3_PATCH_POS (TERMS == 6 ? 2 : 3)
#endif
#ifndef B3_BRANCHFREE
#define B3_BRANCHFREE 1
#endif
#ifndef B3_ZERO_EARLY
#define B3_ZERO_EARLY 1
#endif
#ifndef B3_XTILE
#define B3_XTILE 0
#endif
#ifndef B3_KB
#define B3_KB 4      // tile rows whose exchange reads are in flight together in the epilogue
#endif
#ifndef B3_PHASED
#define B3_PHASED 0
#endif
#if B3_PHASED
#define B3_WAITVM()                                                                               \
    do {                                                                                          \
        __builtin_amdgcn_s_waitcnt(0x0F70);     /* vmcnt(0) */                                    \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    } while (0)
#else
#define B3_WAITVM() do { } while (0)
#endif
// Where in a step the patch loads are issued: 0 = in front of the first fragment loads (one phase until the in-order vmcnt forces them),
// 1 / 2 / 3 = behind the fragment loads of phase 0 / 1 / 2 (two phases)
#define B3_PATCH_AT(POS)                                                                          \
    do {                                                                                          \
        if (B3_PATCH_POS == (POS)) B3_LOADP(pr);                                                  \
    } while (0)
// Order of a step's plane columns.  Column j needs the row-combined patch columns t: 0: t0 - t2, 1: t1 + t2, 2: t2 - t1, 3: t1 - t3, and
// the NEXT step's t replace the current ones as they die.  In the order 0, 3, 1, 2 (shipped) t2 is reloaded at the end of the phase in
// front of its first use; the order 0, 2, 1, 3 gives every reload a whole phase between its LDS reads and its first use -- and measures
// 1 % slower on every layer (same bits; profiles/r06_b3_plane_order_ab.log): the LDS waits (7 % of the wave cycles) are not these.
#if B3_PLANE_ORDER
#define B3_JB 2     // multiplied in phase B (operands produced in A)
#define B3_JD 3     // multiplied in phase D (operands produced in C)
#else
#define B3_JB 3
#define B3_JD 2
#endif
#define B3_STEP(WNEXT)                                                                            \
    do {                                                                                          \
        B3_WAITVM();                                                                              \
        B3_PATCH_AT(0);                                                                           \
        B3_ULOAD(woff, B3_JB, 1);                                                                 \
        B3_PATCH_AT(1);                                                                           \
        B3_VPLANE(B3_JB, 1);                                                                      \
        B3_MFMAS(0, 0);                                                                           \
        B3_TCOL(0, bo1);                                                                          \
        BSTAMP(8);                                                                                \
        B3_WAITVM();                                                                              \
        B3_ULOAD(woff, 1, 0);                                                                     \
        B3_PATCH_AT(2);                                                                           \
        B3_VPLANE(1, 0);                                                                          \
        B3_MFMAS(B3_JB, 1);                                                                       \
        B3_TCOL(B3_JB, bo1);                                                                      \
        BSTAMP(9);                                                                                \
        B3_WAITVM();                                                                              \
        B3_ULOAD(woff, B3_JD, 1);                                                                 \
        B3_PATCH_AT(3);                                                                           \
        B3_VPLANE(B3_JD, 1);                                                                      \
        B3_MFMAS(1, 0);                                                                           \
        B3_TCOL(1, bo1);                                                                          \
        B3_TCOL(B3_JD, bo1);                                                                      \
        BSTAMP(10);                                                                               \
        B3_WAITVM();                                                                              \
        woff = (WNEXT);                                                                           \
        B3_ULOAD(woff, 0, 0);                                                                     \
        B3_VPLANE(0, 0);                                                                          \
        B3_MFMAS(B3_JD, 1);                                                                       \
        BSTAMP(11);                                                                               \
        B3_COMMIT(pr, bo2);                                                                       \
        __syncthreads();                                                                          \
        BSTAMP(12);                                                                               \
        const int tt_ = bo1; bo1 = bo2; bo2 = tt_;                                                \
    } while (0)
// A tile's last step: nothing of the next step is produced (the next tile starts from its staged images after the epilogue, so that no
// operand registers are live across it); only the next tile's first weight fragments are sent for.
#define B3_STEP_LAST(WNEXT)                                                                       \
    do {                                                                                          \
        B3_WAITVM();                                                                              \
        B3_PATCH_AT(0);                                                                           \
        B3_ULOAD(woff, B3_JB, 1);                                                                 \
        B3_PATCH_AT(1);                                                                           \
        B3_VPLANE(B3_JB, 1);                                                                      \
        B3_MFMAS(0, 0);                                                                           \
        B3_WAITVM();                                                                              \
        B3_ULOAD(woff, 1, 0);                                                                     \
        B3_PATCH_AT(2);                                                                           \
        B3_VPLANE(1, 0);                                                                          \
        B3_MFMAS(B3_JB, 1);                                                                       \
        B3_WAITVM();                                                                              \
        B3_ULOAD(woff, B3_JD, 1);                                                                 \
        B3_PATCH_AT(3);                                                                           \
        B3_VPLANE(B3_JD, 1);                                                                      \
        B3_MFMAS(1, 0);                                                                           \
        B3_WAITVM();                                                                              \
        woff = (WNEXT);                                                                           \
        B3_ULOAD(woff, 0, 0);                                                                     \
        B3_MFMAS(B3_JD, 1);                                                                       \
        B3_COMMIT(pr, bo2);                                                                       \
        __syncthreads();                                                                          \
        const int tt_ = bo1; bo1 = bo2; bo2 = tt_;                                                \
    } while (0)

#ifdef PIVLFN_STAMPS
    // tools build: ticks (s_memtime) of wave 0 per category, summed over the workgroup's tiles: 0 tile start, 1 full steps, 2 last step,
    // 3 epilogue, 4 load-stream tile switch, 5 whole, 6 tiles, 7 steps, 8-11 the four phases of the full steps, 12 their closing barrier
    const bool stamp_ = p.stamps != nullptr && wave == 0;
    unsigned long long tk_ = 0, t_begin_ = 0, sd_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (stamp_) t_begin_ = tk_ = __builtin_amdgcn_s_memtime();
#define BSTAMP(I)                                                                                 \
    do {                                                                                          \
        if (stamp_) {                                                                             \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                         \
            sd_[I] += now_ - tk_;                                                                 \
            tk_ = now_;                                                                           \
        }                                                                                         \
    } while (0)
#else
#define BSTAMP(I) do { } while (0)
#endif
    f32x4 pr[BPS];
    f32x4 t[2][4][2];               // [tile block][patch column][quad]: row-combined columns of the current step
    u32x4 V[2][2][3], U[2][2][3];   // [slot][tile block | channel block][piece]
    int bo1 = BBUF_A, bo2 = BBUF_B; // the image read during a step (the NEXT step's patch) and the buffer the step after next is committed to
    B3_LANE_ADDRS();
    B3_LTILE();
    // prologue: the first two patches of the stream and the first weight fragments in flight together
    {
        f32x4 pr2[BPS];
        B3_ULOAD(woff, 0, 0);
        B3_LOADP(pr);
        B3_LOADP(pr2);
        B3_COMMIT(pr, bo2);
        B3_COMMIT(pr2, bo1);
    }
    __syncthreads();

    // p.nchunk >= 2 (launcher).  The load stream is two steps ahead: entering a tile it stands at the tile's third step, and it moves on
    // to the next tile in front of the tile's last but one step.
#if B3_XTILE
    B3_TCOL(0, bo2); B3_TCOL(1, bo2); B3_TCOL(2, bo2); B3_TCOL(3, bo2);
    B3_VPLANE(0, 0);
    __syncthreads();
#endif
    for (int ti = 0; ti < ntiles; ++ti) {
#if !B3_XTILE
        // tile start: bo2 holds the tile's first image, bo1 its second
        B3_LANE_ADDRS();
        B3_TCOL(0, bo2); B3_TCOL(1, bo2); B3_TCOL(2, bo2); B3_TCOL(3, bo2);
        B3_VPLANE(0, 0);
        __syncthreads();            // every wave has read the first image: its buffer is free for the tile's third step
#endif
        BSTAMP(0);
        const bool has_next = ti + 1 < ntiles;
        for (int c = 0; c + 1 < p.nchunk; ++c) {
            if (c == p.nchunk - 2) {
                B3_LNEXT();
                BSTAMP(4);
            }
            B3_STEP(woff + wstep);
#ifdef PIVLFN_STAMPS
            if (stamp_) { const unsigned long long ph_ = sd_[8] + sd_[9] + sd_[10] + sd_[11] + sd_[12]; sd_[1] = ph_; sd_[7] += 1; }
#endif
        }
        // the step after a tile's last one is the next tile's first: the load stream stands on that tile (past the last tile: the same
        // fragments again, loaded and never used)
        const int wfirst = has_next ? (lng * 4 + wave) * BWAVE * 16 : woff;
#if B3_XTILE
        B3_STEP(wfirst);
#else
        B3_STEP_LAST(wfirst);
#endif
        BSTAMP(2);
        __builtin_amdgcn_sched_barrier(0);      // nothing of the epilogue is hoisted into the step (the first planes' accumulators are final early: their reads would be)

        // ---- output transform.  acc[jp][mb][nw][4 rg + e] = M[(wave, jp)][cout 64 ng + 32 nw + 8 rg + 4 g + e][tile n of block mb]
        // column half (in registers): R[0] = M0 + M1 + M2, R[1] = M1 - M2 - M3; row half across waves: Y[0] = R_0 + R_1 + R_2, Y[1] = R_1 - R_2 - R_3.
        // Exchange layout per (nw, wave i, q): 256 quads, position of (tile n, channel quad c4 = 2 rg + g of the 32-channel block) =
        // (n >> 3) * 64 + (n & 7) * 8 + (c4 ^ (n & 7)): the finishing wave reads 64 CONSECUTIVE quads per instruction (tile row k: 8 tiles x
        // 8 channel quads), so that eight lanes hold one pixel's 32 channels and every store instruction writes eight whole 128-byte
        // lines (a lane per tile and 16 bytes of it -- 64 lines per instruction -- took 4100 cycles per half); the XOR keeps the
        // writers' eight-lane groups on distinct banks.
        f32x4 *xch = smem4 + BXCH;
        const int ng = cng, x0 = cx0, y0 = cy0, b = cb;
        const int pp = wave >> 1, qq = wave & 1;      // this wave finishes output pixel (pp, qq) of every tile
        // (the lane id is laundered per tile: left visible, the epilogue's per-lane addresses are loop invariants that the compiler keeps in
        //  registers through the K steps -- and spills; a spill reload waits vmcnt(0))
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int en = ln & 31, eg = ln >> 5;
        const int wpos = (en >> 3) * 64 + (en & 7) * 8;
        const int rts = ln >> 3, rc4 = (ln & 7) ^ rts;     // finishing lane: tile column, channel quad
        // every load of the epilogue in front of its first store (a load behind a store waits for the store: vmcnt counts both in order)
        f32x4 bias4[2];
#pragma unroll
        for (int nw = 0; nw < 2; ++nw) bias4[nw] = *reinterpret_cast<const f32x4 *>(p.bias + (ng * 2 + nw) * 32 + 4 * rc4);
        // stores through a descriptor that starts at the tile's first image row; pixels outside the image and channels past cout_store get
        // an out-of-range offset instead of a branch
        const size_t orow = ((size_t)b * p.H + y0) * p.W;
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(p.out + orow * p.out_stride, 0,
            (unsigned)min(((size_t)(p.H - y0) * p.W) * p.out_stride * 4, (size_t)0x7fffffff), 0x00020000);
        const int ox = x0 + 2 * rts + qq;
#ifdef B3_ABL_NOEPI      // timing build: no output transform at all
        if (p.H < 0)
#endif
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            if (mb) __syncthreads();            // the first half has been read
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nw = 0; nw < 2; ++nw)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    f32x4 m[4];
#pragma unroll
                    for (int jp = 0; jp < 4; ++jp)
                        m[jp] = f32x4{acc[jp][mb][nw][4 * rg + 0], acc[jp][mb][nw][4 * rg + 1], acc[jp][mb][nw][4 * rg + 2], acc[jp][mb][nw][4 * rg + 3]};
                    const int wq = wpos + ((2 * rg + eg) ^ (en & 7));
                    xch[((nw * 4 + wave) * 2 + 0) * 256 + wq] = (m[0] + m[1]) + m[2];
                    xch[((nw * 4 + wave) * 2 + 1) * 256 + wq] = (m[1] - m[2]) - m[3];
                }
#if B3_ZERO_EARLY
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][mb][nb][r] = 0.f;       // this half's accumulators are read: zero them under the writes' latency
#endif
            __syncthreads();
            BSTAMP(13 + 0 * mb);
            // branch-free (the row pair is a wave-uniform offset and a sign), B3_KB tile rows' reads in flight together: with one wave per
            // SIMD every LDS round trip that is waited for on its own is exposed
            const float sg = pp ? -1.f : 1.f;
#pragma unroll
            for (int nw = 0; nw < 2; ++nw) {
                const int ch = (ng * 2 + nw) * 32 + 4 * rc4;
#pragma unroll
                for (int k0 = 0; k0 < 4; k0 += B3_KB) {
                    f32x4 xv[B3_KB][3];
#pragma unroll
                    for (int kk = 0; kk < B3_KB; ++kk)
#pragma unroll
                        for (int i = 0; i < 3; ++i) xv[kk][i] = xch[((nw * 4 + pp + i) * 2 + qq) * 256 + (k0 + kk) * 64 + ln];
#pragma unroll
                    for (int kk = 0; kk < B3_KB; ++kk) {
                        const int k = k0 + kk;
                        f32x4 y;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            // pp = 0: (R_0 + R_1) + R_2;  pp = 1: (R_1 - R_2) - R_3  (sg * x is exact)
                            const float v = __builtin_fmaf(sg, xv[kk][2][e], __builtin_fmaf(sg, xv[kk][1][e], xv[kk][0][e])) + bias4[nw][e];
                            y[e] = p.lrelu ? lrelu01(v) : v;
                        }
                        const int oyl = 2 * (mb * 4 + k) + pp;
                        const bool ok = y0 + oyl < p.H && ox < p.W && ch < p.cout_store;
#ifdef B3_ABL_NOSTORE      // timing build: every store out of range (dropped by the range check)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), ors, (int)(ok && p.H < 0 ? (unsigned)((oyl * p.W + ox) * p.out_stride + ch) * 4u : WOOB), 0, 0);
#else
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), ors, (int)(ok ? (unsigned)((oyl * p.W + ox) * p.out_stride + ch) * 4u : WOOB), 0, 0);
#endif
                    }
                }
            }
            BSTAMP(14);
        }
#if !B3_ZERO_EARLY
        B3_ZERO_ACC();
#endif
        BSTAMP(15);
        BSTAMP(3);
#ifdef PIVLFN_STAMPS
        if (stamp_) sd_[6] += 1;
#endif
    }
#ifdef PIVLFN_STAMPS
    if (stamp_ && lane == 0) {
        sd_[5] = __builtin_amdgcn_s_memtime() - t_begin_;
        for (int i = 0; i < 16; ++i) p.stamps[(size_t)blockIdx.x * 16 + i] = sd_[i];
    }
#endif
#undef BSTAMP
#undef B3_PVO
#undef B3_LANE_ADDRS
#undef B3_LTILE
#undef B3_LOADP
#undef B3_COMMIT
#undef B3_WAITVM
#undef B3_LNEXT
#undef B3_TCOL
#undef B3_VPLANE
#undef B3_ULOAD
#undef B3_ONE
#undef B3_ZERO_ACC
#undef B3_MFMAS
#undef B3_STEP
#undef B3_STEP_LAST
}

// fp32 -> bf16 bits, round to nearest even (finite inputs)
static unsigned short bf16_rne(float f)
{
    unsigned u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float bf16_val(unsigned short h)
{
    const unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// OIHW [cout][cin][3][3] -> Winograd-domain weights, each split into three bf16 pieces, in MFMA A-fragment order:
//   [step][64-channel group][plane row i][plane col j][channel block 2][piece 3][lane 64][8]: lane = (cout & 31) + 32 * k-half, element e
//   multiplies staged channel 16 * step_in_source + 8 * k-half + e of the step's source.  U = G g G^T in float64, rounded once to fp32
//   (the same U conv_wino.hip multiplies by), then u = h + m + l exactly.
void pack_conv_wb(const float *w, int cout, int cin, const int *creal, const int *cload, const int *coff, int nseg,
                  std::vector<unsigned short> &pk, int *nstep_out)
{
    const int cp = (cout + 63) / 64 * 64, NG = cp / 64;
    int nstep = 0;
    for (int s = 0; s < nseg; ++s) nstep += (cload[s] + 15) / 16;
    const int nreal = nstep;
    nstep = std::max(nstep, 2);      // the kernel's pipeline wants two steps per tile: a 16-channel layer gets a second, all-zero one
    (void)nreal;
    pk.assign((size_t)nstep * NG * 4 * BWAVE * 8, 0);
    static const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    int step = 0, run = 0;
    for (int s = 0; s < nseg; ++s) {
        const int off = coff[s] >= 0 ? coff[s] : run;
        for (int c0 = 0; c0 < cload[s]; c0 += 16, ++step)
            for (int kh = 0; kh < 2; ++kh)
                for (int e = 0; e < 8; ++e) {
                    const int c = c0 + 8 * kh + e;
                    if (c >= creal[s]) continue;
                    for (int o = 0; o < cout; ++o) {
                        const float *gk = w + ((size_t)o * cin + off + c) * 9;
                        double tmp[4][3];
                        for (int i = 0; i < 4; ++i)
                            for (int x = 0; x < 3; ++x) tmp[i][x] = G[i][0] * gk[0 * 3 + x] + G[i][1] * gk[1 * 3 + x] + G[i][2] * gk[2 * 3 + x];
                        const int grp = o >> 6, nb = (o >> 5) & 1, ln = (o & 31) + 32 * kh;
                        for (int i = 0; i < 4; ++i)
                            for (int j = 0; j < 4; ++j) {
                                const float u = (float)(tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2]);
                                const unsigned short h = bf16_rne(u);
                                const float r1 = u - bf16_val(h);
                                const unsigned short m = bf16_rne(r1);
                                const float r2 = r1 - bf16_val(m);
                                const unsigned short l = bf16_rne(r2);
                                const size_t base = ((((((size_t)step * NG + grp) * 4 + i) * 4 + j) * 2 + nb) * 3) * 64 * 8;
                                pk[base + (0 * 64 + ln) * 8 + e] = h;
                                pk[base + (1 * 64 + ln) * 8 + e] = m;
                                pk[base + (2 * 64 + ln) * 8 + e] = l;
                            }
                    }
                }
        run += creal[s];
    }
    *nstep_out = nstep;
}

// The layers this kernel takes: whole 64-channel groups (the 32- and 96-channel layers stay on conv_wino.hip)
bool conv_wino_b3_supports(int cout_pad) { return cout_pad % 64 == 0; }

template <int TERMS>
static int launch_b3(const ConvParamsW &p, hipStream_t st)
{
    const size_t lds = (size_t)BLDS_QUADS * 16;
    static LdsAttr attr;
    if (int rc = ensure_dyn_lds(attr, reinterpret_cast<const void *>(conv_wino_b3_kernel<TERMS>), (int)lds)) return rc;
    const long tiles = (long)p.B * cdiv(p.H, 16) * cdiv(p.W, 16) * (p.cout_pad / 64);
    PIV_REQUIRE(tiles < (1L << 31), "conv_wino_b3: too many tiles");
    const long blocks = std::min<long>(tiles, device_cus());      // persistent: one workgroup per CU walks its share of the tiles
    hipLaunchKernelGGL((conv_wino_b3_kernel<TERMS>), dim3((unsigned)blocks), dim3(256), lds, st, p);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

// p.wpk_b = pack_conv_wb's array, p.nchunk = its K steps of 16 channels, p.terms = 6, 8 or 9 piece products per product
int launch_conv_wb(const ConvParamsW &p_in, hipStream_t st)
{
    ConvParamsW p = p_in;
    p.stamps = reinterpret_cast<unsigned long long *>(((unsigned long long)(unsigned)PIV_KNOB(6) << 32) | (unsigned)PIV_KNOB(5));   // tools only (0 in production)
    PIV_REQUIRE(p.nseg >= 1 && p.nseg <= 3 && p.wpk_b && p.bias && p.out, "conv_wino_b3: bad arguments");
    PIV_REQUIRE(p.cout_pad % 64 == 0 && p.cout_store <= p.cout_pad && p.cout_store % 4 == 0 && p.out_stride % 4 == 0,
                "conv_wino_b3: cout_pad=%d cout_store=%d out_stride=%d", p.cout_pad, p.cout_store, p.out_stride);
    PIV_REQUIRE(p.B > 0 && p.H > 0 && p.W > 0 && (long)p.B * p.H * p.W < (1L << 31), "conv_wino_b3: bad shape");
    for (int s = 0; s < p.nseg; ++s)      // 32-bit byte offsets inside the rows of one patch (descriptors are rebased per workgroup)
        PIV_REQUIRE((long)20 * p.W * p.seg[s].stride * 4 < (1L << 31), "conv_wino_b3: 20 rows of source %d exceed 2 GiB", s);
    PIV_REQUIRE((long)16 * p.W * p.out_stride * 4 < (1L << 31), "conv_wino_b3: 16 output rows exceed 2 GiB");
    int nstep = 0;
    for (int s = 0; s < p.nseg; ++s) {
        PIV_REQUIRE(p.seg[s].cload % 4 == 0 && p.seg[s].stride % 4 == 0 && p.seg[s].ptr, "conv_wino_b3: segment %d misaligned", s);
        nstep += (p.seg[s].cload + 15) / 16;
    }
    nstep = std::max(nstep, 2);
    PIV_REQUIRE(nstep == p.nchunk, "conv_wino_b3: segments hold %d steps, weights were packed for %d", nstep, p.nchunk);
    PIV_REQUIRE((size_t)p.nchunk * (p.cout_pad / 64) * 4 * BWAVE * 16 < ((size_t)1 << 32), "conv_wino_b3: packed weights exceed 4 GiB");
    switch (p.terms) {
        case 6: return launch_b3<6>(p, st);
        case 8: return launch_b3<8>(p, st);
        case 9: return launch_b3<9>(p, st);
        default: PIV_REQUIRE(false, "conv_wino_b3: terms=%d (6, 8 or 9)", p.terms);
    }
    return PIVLFN_OK;
}

}  // namespace pivlfn
