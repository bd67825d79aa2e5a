// Winograd F(2x2, 3x3) on persistent workgroups with specialised waves: four waves that transform and multiply, four that move data.
//
// Same layers, same packed weights, same arithmetic and the same summation order per output value as conv_wino.hip (chunks
// ascending, k = {j, 4 + j} inside a chunk, the output transform's additions in the same order): the results are bit-identical to
// that kernel's, so which of the two runs a layer may follow the launch size.  What changes is who issues what, and when.
//
// In conv_wino.hip every wave loads its share of the patch, commits it to LDS, transforms, multiplies and stores, and a workgroup
// lives for one tile: of a launch's time 9 % is the patch traffic issued between the MFMAs and 11-30 % the workgroup's prologue
// and epilogue.  Here:
//   * One workgroup per CU (persistent, 768 threads = three waves per SIMD).  Waves 0-7 -- two per SIMD -- are CONSUMERS: consumers i
//     and 4 + i share plane row i of a 32-tile x 32 NBW-channel work item, planes (i, 0..1) and (i, 2..3).  A consumer's K step is a
//     wave's of conv_wino.hip without the staging: 8 NBW MFMAs, the raw operand reads of the NEXT chunk from LDS (two patch rows x
//     three columns), its share of the transform (10 packed instructions) and the weight fragments of the next chunk (buffer loads
//     with scalar offsets behind each plane's MFMAs, into the registers that plane just read).  Two such waves per SIMD keep the
//     matrix pipe fed the way two co-resident workgroups of conv_wino.hip do in their K loops: one wave's loads, waits and vector
//     instructions are issued under the other's MFMAs (ONE consumer per SIMD pays every non-MFMA instruction in matrix time:
//     round 5's second version, 0.68 of the peak).
//   * Waves 8-11 -- one per SIMD -- are PRODUCERS: global -> registers -> LDS (two steps of latency cover, two register sets)
//     for the raw-patch ring, and the stores.  Their steady state contains NO vector instruction: on this chip a vector instruction
//     of the second wave of a SIMD is not executed beside a wave that streams fp32 MFMAs and meets it at barriers -- it waits until
//     that wave pauses (tools/micro/ws_step.hip, ws_gap.hip: LDS writes, scalar work and loads of the second wave are free, its
//     vector work adds its whole duration to the step; round 5's first version, whose producers also transformed, ran at 0.60 of
//     the matrix peak for that reason).  So the per-lane load offsets are constants (interior items: the descriptor starts at the
//     patch's first pixel) chosen by scalar branches, the LDS addresses are registers plus immediates, and everything that needs the
//     vector unit -- the offsets of an item that touches the image border, and the row half of the output transform with bias and
//     LeakyReLU -- happens ONCE per item, in the step where the consumers wait for it anyway.
//   * The work items of a workgroup (tile x channel group; its XCD's band, interleaved over the XCD's CUs) form ONE step sequence:
//     the producers' loads run four chunks ahead and cross item boundaries, so a consumer goes from the last MFMA of an item to the
//     first of the next with only the column half of the output transform in between (it writes R = M A to LDS).
//   * One s_barrier per K step hands the raw ring over: at the barrier that ends step c, raw(c + 2) is in LDS and raw(c)'s slot free.
#include <algorithm>
#include "common.h"
#include "wino_common.h"

namespace pivlfn {

namespace {

constexpr int WS_PH = 10;                              // patch rows of an 8 x 4 block of 2x2 tiles
constexpr int WS_NSLOT = WS_PH * WPW * 2;              // 16-byte staging slots of one chunk's patch (360)
constexpr int WS_PS = 2;                               // slots per producer thread
constexpr int WS_PBUF = WS_PH * WROWQ + WPIXQ;         // quads per raw patch buffer (+ one spare record)
constexpr int WS_XQ = 1152;                            // quad offset of the output-transform exchange (NBW x 4096 quads)
static_assert(2 * WS_PBUF <= WS_XQ, "raw ring overlaps the exchange area");

}  // namespace

template <int NBW>
__global__ __launch_bounds__(768) void conv_wino_ws_kernel(const ConvParamsW p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4 *smem4 = reinterpret_cast<f32x4 *>(smem);

    const int NB = p.cout_pad >> 5;             // 32-channel blocks of the layer
    const int NG = NB / NBW;                    // channel groups per spatial tile
    const int tiles_x = (p.W + 15) >> 4, tiles_y = (p.H + 7) >> 3;
    const int nchunk = p.nchunk;
    // Work items of this workgroup.  Blocks b and b + 8 share an XCD: XCD x takes a contiguous band of items (the channel groups
    // of a spatial tile are consecutive items, then the tile's neighbours along x), and its K workgroups walk the band interleaved,
    // so at any time an XCD works on K consecutive items and a patch is fetched from HBM once.
    int t0, tstep, n_items;
    {
        const int N = p.B * tiles_y * tiles_x * NG;
        const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3, K = gridDim.x >> 3;
        const int q = N >> 3, r = N & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int cnt = q + (xcd < r ? 1 : 0);
        if (kk >= cnt) return;
        n_items = (cnt - kk + K - 1) / K;
        t0 = base + kk;
        tstep = K;
    }

    const int tid = threadIdx.x;
    // the layer's bias lives in LDS (behind the exchange area) for the life of the workgroup: the item's last step must not wait on
    // a global load that queues behind the patch loads in flight
    const int bias_q = WS_XQ + NBW * 4096;
    for (int i = tid; i < p.cout_pad; i += 768) smem[bias_q * 4 + i] = p.bias[i];
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, g = lane >> 5;
    const int tyl = n >> 3, txl = n & 7;
    float m1 = -1.f;
    asm("" : "+v"(m1));
#ifdef PIVLFN_STAMPS
    // tools build: ablation mask p.dbg (1 no patch loads, 2 no weight refills, 4 no raw reads, 8 no finish_item, 16 no column half,
    // 32 no transform) and per-workgroup stamps of wave 0 (consumer) and wave 4 (producer): ticks spent waiting at the step barriers
    const int dbg = p.dbg;
    const bool stamp_ = p.stamps != nullptr && (wave == 0 || wave == 8) && blockIdx.x < 4096;
    unsigned long long tk_ = 0, t_begin_ = 0, d_bar_ = 0, d_a_ = 0, d_b_ = 0;
    if (stamp_) t_begin_ = tk_ = __builtin_amdgcn_s_memtime();
#define WS_STAMP(ACC)                                                                             \
    do {                                                                                          \
        if (stamp_) {                                                                             \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                         \
            ACC += now_ - tk_;                                                                    \
            tk_ = now_;                                                                           \
        }                                                                                         \
    } while (0)
#define WS_DBG(M) (dbg & (M))
#else
#define WS_STAMP(ACC) do { } while (0)
#define WS_DBG(M) 0
#endif
#define WS_SYNC()                                                                                 \
    do {                                                                                          \
        WS_STAMP(d_a_);                                                                           \
        __syncthreads();                                                                          \
        WS_STAMP(d_bar_);                                                                         \
    } while (0)
    auto decode = [&](int s, int &b, int &y0, int &x0, int &nb0) {
        int t = t0 + s * tstep;
        nb0 = (t % NG) * NBW;
        t /= NG;
        x0 = (t % tiles_x) * 16;
        t /= tiles_x;
        y0 = (t % tiles_y) * 8;
        b = t / tiles_y;
    };

    if (wave < 8) {
        // ------------------------------------------------------------------------------------ consumer: planes (pi, 2 ph) and (pi, 2 ph + 1)
        const int pi = wave & 3, ph = wave >> 2;
        f32x16 acc[2][NBW];
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x4 V[2], Wt[NBW][2], rawa[3] = {}, rawb[3] = {}, tt[3];
        // weights of (chunk, block nb, plane row i): 4 planes x 64 lanes x 16 bytes, contiguous: byte offset ((chunk NB + nb) 4 + i) 4096
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.wpk), 0,
                                                                            (unsigned)((size_t)nchunk * NB * 16 * 1024), 0x00020000);
        const int wvoff = lane * 16 + ph * 2048;      // the wave's first plane inside the plane row's 4 KB
        const int wstep = NB * 4 * 4096;              // bytes from one chunk to the next
        // plane row i = pi: (B^T d)[i][.] = d[ra][.] + sb * d[rb][.]; this wave needs patch columns ph, ph + 1, ph + 2
        const int ra = pi == 0 ? 0 : (pi == 2 ? 2 : 1);
        const int rb = pi == 0 ? 2 : (pi == 1 ? 2 : (pi == 2 ? 1 : 3));
        const float sb = pi == 1 ? 1.f : -1.f;
        int aq[3], bq[3];                             // quad indices of the lane's six operands in raw slot 0
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int c = ph + k;
            const int coff = ((c >> 1) + (c & 1) * 9) * WPIXQ;
            aq[k] = (tyl + (ra >> 1) + (ra & 1) * (WS_PH / 2)) * WROWQ + txl * WPIXQ + g + coff;      // patch pixel (2 tyl + ra, 2 txl + c)
            bq[k] = (tyl + (rb >> 1) + (rb & 1) * (WS_PH / 2)) * WROWQ + txl * WPIXQ + g + coff;
        }

#define WS_MFMA(JP, Z)                                                                            \
    do {                                                                                          \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                             \
            _Pragma("unroll") for (int nw = 0; nw < NBW; ++nw)                                    \
                acc[JP][nw] = __builtin_amdgcn_mfma_f32_32x32x2f32(Wt[nw][JP][j], V[JP][j], ((Z) && j == 0) ? zero16 : acc[JP][nw], 0, 0, 0); \
    } while (0)
#define WS_REFILLW(JP, SOFF)                                                                      \
    do {                                                                                          \
        if (!WS_DBG(2)) _Pragma("unroll") for (int nw = 0; nw < NBW; ++nw)                        \
            Wt[nw][JP] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wvoff + (JP)*1024, (SOFF) + nw * 16384, 0)); \
    } while (0)
// the lane's 6 operand quads (rows ra, rb x three columns of its tile's patch) from the raw patch at quad offset RQ
#define WS_READRAW(RQ)                                                                            \
    do {                                                                                          \
        if (!WS_DBG(4)) _Pragma("unroll") for (int k = 0; k < 3; ++k) {                           \
            rawa[k] = smem4[(RQ) + aq[k]];                                                        \
            rawb[k] = smem4[(RQ) + bq[k]];                                                        \
        }                                                                                         \
    } while (0)
// V = (B^T d B)[pi][2 ph .. 2 ph + 1] in place: row combination of the three columns, then planes 0, 1 from columns 0-2 or planes
// 2, 3 from columns 1-3 (the same instructions on the same values as conv_wino.hip's WINO_XFORM)
#define WS_XFORM()                                                                                \
    do {                                                                                          \
        _Pragma("unroll") for (int k = 0; k < 3; ++k) tt[k] = sub4(rawa[k], rawb[k], sb);         \
        if (ph == 0) { V[0] = sub4(tt[0], tt[2], m1); V[1] = tt[1] + tt[2]; }                     \
        else { V[0] = sub4(tt[1], tt[0], m1); V[1] = sub4(tt[0], tt[2], m1); }                    \
    } while (0)
// One K step: chunk c's MFMAs, plane by plane; behind a plane's MFMAs the refill of its weight registers with chunk c + 1; the raw
// reads of chunk c + 1 behind the first plane, its transform (in place: a plane's operand is dead once its MFMAs are issued) last.
#define WS_CSTEP(Z, RQ, SOFF)                                                                     \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        WS_MFMA(0, Z);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        WS_REFILLW(0, SOFF);                                                                      \
        WS_READRAW(RQ);                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        WS_MFMA(1, Z);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        WS_REFILLW(1, SOFF);                                                                      \
        if (!WS_DBG(32)) WS_XFORM();                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    } while (0)

        auto item_woff = [&](int s) {
            const int t = t0 + s * tstep;
            return (((t % NG) * NBW) * 4 + pi) * 4096;
        };
        int woff = item_woff(0);
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) WS_REFILLW(jp, woff);
        WS_SYNC();      // step -2: raw(0) committed
        WS_READRAW(0);
        WS_XFORM();
        WS_SYNC();      // step -1: raw(1) committed
        int rq = WS_PBUF;     // quad offset of the raw slot the next step reads (chunk c + 1)
        for (int s = 0; s < n_items; ++s) {
            const bool more = s + 1 < n_items;
            const int woff_next = more ? item_woff(s + 1) : 0;
            int soff = woff + wstep;
            WS_CSTEP(true, rq, soff);
            WS_SYNC();
            rq ^= WS_PBUF;
            for (int c = 1; c + 1 < nchunk; ++c) {
                soff += wstep;
                WS_CSTEP(false, rq, soff);
                WS_SYNC();
                rq ^= WS_PBUF;
            }
            // last chunk: the refills and the transform are those of the next item's first chunk (past the end: nobody uses them)
            soff = more ? woff_next : soff;
            WS_CSTEP(false, rq, soff);
            // column half of the output transform, R[0] = (M0 + M1) + M2 and R[1] = (M1 - M2) - M3, split over the two waves of the
            // plane row: this wave leaves (M0 + M1, M1) or (M2, M3) in LDS [nw][plane row][ph][2][rg][lane]; the producers finish it
            f32x4 *xch = smem4 + WS_XQ;
            WS_STAMP(d_a_);
            if (!WS_DBG(16))
#pragma unroll
            for (int nw = 0; nw < NBW; ++nw)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    f32x4 m[2];
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp)
                        m[jp] = f32x4{acc[jp][nw][4 * rg + 0], acc[jp][nw][4 * rg + 1], acc[jp][nw][4 * rg + 2], acc[jp][nw][4 * rg + 3]};
                    xch[((((nw * 4 + pi) * 2 + ph) * 2 + 0) * 4 + rg) * 64 + lane] = ph == 0 ? m[0] + m[1] : m[0];
                    xch[((((nw * 4 + pi) * 2 + ph) * 2 + 1) * 4 + rg) * 64 + lane] = m[1];
                }
            WS_STAMP(d_b_);
            WS_SYNC();
            rq ^= WS_PBUF;
            woff = woff_next;
        }
#ifdef PIVLFN_STAMPS
        if (stamp_ && lane == 0) {
            unsigned long long *o = p.stamps + (size_t)blockIdx.x * 16;
            o[0] = d_a_; o[1] = d_b_; o[2] = d_bar_; o[3] = __builtin_amdgcn_s_memtime() - t_begin_; o[4] = t_begin_; o[5] = (unsigned long long)n_items;
        }
#endif
#undef WS_CSTEP
#undef WS_MFMA
#undef WS_REFILLW
#undef WS_READRAW
#undef WS_XFORM
        return;
    }

    // ---------------------------------------------------------------------------------------------------- producer
    const int pw = wave - 8;
    const int ptid = tid - 512;
    // staging slots of this thread: slot k covers (pixel, quad) = (idx >> 1, idx & 1), idx = ptid + 256 k (conv_wino.hip)
    int plds[WS_PS], ppy[WS_PS], ppx[WS_PS];
    const int q4 = (ptid & 1) * 4;
#pragma unroll
    for (int k = 0; k < WS_PS; ++k) {
        const int idx = ptid + 256 * k;
        const int pix = idx >> 1;
        const int py = pix / WPW, px = pix - py * WPW;
        ppy[k] = idx < WS_NSLOT ? py : -100000;
        ppx[k] = px;
        plds[k] = (idx < WS_NSLOT ? ((py >> 1) + (py & 1) * (WS_PH / 2)) * WROWQ + ((px >> 1) + (px & 1) * 9) * WPIXQ : WS_PH * WROWQ) + (ptid & 1);
    }

    // ---- load cursor: the (item, chunk) whose raw patch is fetched next.  Everything a load needs is scalar -- descriptor, channel
    // offset -- or a register computed once: per-lane byte offsets of the slot's pixel inside an INTERIOR item's patch, whose descriptor
    // starts at the patch's first pixel (iv*), and, for an item that touches the image border, offsets from the first image row of
    // its patch with out-of-range values for the zero padding (bv*: vector work, done when the cursor enters such an item).  `*t` are
    // the versions for a source's 4-channel tail chunk (the upper quad loads zeros).
    const size_t img_px = (size_t)p.H * p.W;
    int l_item = 0, l_chunk = 0, l_seg = 0, l_c0 = 0;
    const int scl0 = p.seg[0].cload, scl1 = p.seg[p.nseg > 1 ? 1 : 0].cload, scl2 = p.seg[p.nseg > 2 ? 2 : 0].cload;
    const int sst40 = p.seg[0].stride * 4, sst41 = p.seg[p.nseg > 1 ? 1 : 0].stride * 4, sst42 = p.seg[p.nseg > 2 ? 2 : 0].stride * 4;
    __amdgpu_buffer_rsrc_t rs0, rs1, rs2;
    bool l_border = false;
    unsigned iv0[WS_PS], iv1[WS_PS], iv2[WS_PS], iv0t[WS_PS], iv1t[WS_PS], iv2t[WS_PS];
    unsigned bv0[WS_PS] = {}, bv1[WS_PS] = {}, bv2[WS_PS] = {}, bv0t[WS_PS] = {}, bv1t[WS_PS] = {}, bv2t[WS_PS] = {};
#pragma unroll
    for (int k = 0; k < WS_PS; ++k) {
        const unsigned pix = ppy[k] >= 0 ? (unsigned)(ppy[k] * p.W + ppx[k]) : WOOB;
        iv0[k] = pix != WOOB ? pix * (unsigned)sst40 + (unsigned)q4 * 4u : WOOB;
        iv1[k] = pix != WOOB ? pix * (unsigned)sst41 + (unsigned)q4 * 4u : WOOB;
        iv2[k] = pix != WOOB ? pix * (unsigned)sst42 + (unsigned)q4 * 4u : WOOB;
        iv0t[k] = q4 ? WOOB : iv0[k];
        iv1t[k] = q4 ? WOOB : iv1[k];
        iv2t[k] = q4 ? WOOB : iv2[k];
    }
    // descriptor of source SS starting at pixel (B_, ROW, COL): 64-bit scalar arithmetic; it ends with the image
#define WS_MAKE_RS(SS, B_, ROW, COL)                                                              \
    __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.seg[SS].ptr + ((size_t)(B_)*img_px + (size_t)(ROW)*p.W + (COL)) * p.seg[SS].stride), 0, \
                                      (unsigned)min((((size_t)(p.H - (ROW)) * p.W - (COL) - 1) * p.seg[SS].stride + p.seg[SS].cload) * 4, (size_t)0x7fffffff), 0x00020000)
#define WS_ENTER_ITEM(S_)                                                                         \
    do {                                                                                          \
        int b_, y0_, x0_, nb0_;                                                                   \
        decode(S_, b_, y0_, x0_, nb0_);                                                           \
        l_border = y0_ < 1 || x0_ < 1 || y0_ + 9 > p.H || x0_ + 17 > p.W;                        \
        const int row0_ = l_border ? max(y0_ - 1, 0) : y0_ - 1, col0_ = l_border ? 0 : x0_ - 1;   \
        rs0 = WS_MAKE_RS(0, b_, row0_, col0_);                                                    \
        rs1 = WS_MAKE_RS(p.nseg > 1 ? 1 : 0, b_, row0_, col0_);                                   \
        rs2 = WS_MAKE_RS(p.nseg > 2 ? 2 : 0, b_, row0_, col0_);                                   \
        if (l_border) {                                                                           \
            _Pragma("unroll") for (int k = 0; k < WS_PS; ++k) {                                   \
                const int iy = y0_ - 1 + ppy[k], ix = x0_ - 1 + ppx[k];                           \
                const bool in = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;                       \
                const unsigned pix = (unsigned)((iy - row0_) * p.W + ix);                         \
                bv0[k] = in ? pix * (unsigned)sst40 + (unsigned)q4 * 4u : WOOB;                   \
                bv1[k] = in ? pix * (unsigned)sst41 + (unsigned)q4 * 4u : WOOB;                   \
                bv2[k] = in ? pix * (unsigned)sst42 + (unsigned)q4 * 4u : WOOB;                   \
                bv0t[k] = q4 ? WOOB : bv0[k];                                                     \
                bv1t[k] = q4 ? WOOB : bv1[k];                                                     \
                bv2t[k] = q4 ? WOOB : bv2[k];                                                     \
            }                                                                                     \
        }                                                                                         \
    } while (0)
// next chunk, next source, next item; past the last item the last chunk is fetched again (into a buffer nobody reads).  Scalar,
// except for the vector work of WS_ENTER_ITEM on a border item.
#define WS_ADVANCE()                                                                              \
    do {                                                                                          \
        const int sclc_ = l_seg == 0 ? scl0 : (l_seg == 1 ? scl1 : scl2);                         \
        if (l_chunk + 1 < nchunk) {                                                               \
            ++l_chunk;                                                                            \
            l_c0 += 8;                                                                            \
            if (l_c0 >= sclc_) {                                                                  \
                ++l_seg;                                                                          \
                l_c0 = 0;                                                                         \
            }                                                                                     \
        } else if (l_item + 1 < n_items) {                                                        \
            ++l_item;                                                                             \
            l_chunk = 0; l_seg = 0; l_c0 = 0;                                                     \
            WS_ENTER_ITEM(l_item);                                                                \
        }                                                                                         \
    } while (0)
    f32x4 prA[WS_PS] = {}, prB[WS_PS] = {};
#define WS_LOAD2(PR, RS, VO)                                                                      \
    do {                                                                                          \
        _Pragma("unroll") for (int k = 0; k < WS_PS; ++k)                                         \
            PR[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(RS, (int)VO[k], l_c0 * 4, 0)); \
    } while (0)
// the cursor's chunk -> PR: the register set of offsets is picked by scalar branches (source, tail chunk, border item)
#define WS_LOADP(PR)                                                                              \
    do {                                                                                          \
        if (!WS_DBG(1)) {                                                                         \
            const int scl_ = l_seg == 0 ? scl0 : (l_seg == 1 ? scl1 : scl2);                      \
            const bool tail_ = l_c0 + 4 >= scl_;                                                  \
            if (!l_border) {                                                                      \
                if (l_seg == 0) { if (tail_) WS_LOAD2(PR, rs0, iv0t); else WS_LOAD2(PR, rs0, iv0); } \
                else if (l_seg == 1) { if (tail_) WS_LOAD2(PR, rs1, iv1t); else WS_LOAD2(PR, rs1, iv1); } \
                else { if (tail_) WS_LOAD2(PR, rs2, iv2t); else WS_LOAD2(PR, rs2, iv2); }         \
            } else {                                                                              \
                if (l_seg == 0) { if (tail_) WS_LOAD2(PR, rs0, bv0t); else WS_LOAD2(PR, rs0, bv0); } \
                else if (l_seg == 1) { if (tail_) WS_LOAD2(PR, rs1, bv1t); else WS_LOAD2(PR, rs1, bv1); } \
                else { if (tail_) WS_LOAD2(PR, rs2, bv2t); else WS_LOAD2(PR, rs2, bv2); }         \
            }                                                                                     \
        }                                                                                         \
    } while (0)
#define WS_COMMIT(PR, RSLOT)                                                                      \
    do {                                                                                          \
        _Pragma("unroll") for (int k = 0; k < WS_PS; ++k) smem4[(RSLOT)*WS_PBUF + plds[k]] = PR[k]; \
    } while (0)

    // The rest of the output transform of item s, whose planes are in the exchange area: producer wave pw finishes output pixel
    // (pp, qq) of every tile -- column half R[qq] of three plane rows from the consumers' halves, row half, bias, LeakyReLU, stores.
    // All LDS reads of a channel block are issued first (no vector instruction in front of them: they travel while the consumers
    // still multiply); the arithmetic runs when the consumers wait at the step's barrier.
    auto finish_item = [&](int s) {
        int b, y0, x0, nb0;
        decode(s, b, y0, x0, nb0);
        const int pp = pw >> 1, qq = pw & 1;
        const f32x4 *xch = smem4 + WS_XQ + lane;
        const int oy = y0 + 2 * tyl + pp, ox = x0 + 2 * txl + qq;
        const bool ok = oy < p.H && ox < p.W;
#pragma unroll
        for (int nw = 0; nw < NBW; ++nw) {
            const int cb = (nb0 + nw) * 32 + 4 * g;
            float *orow = p.out + (size_t)((b * p.H + (ok ? oy : 0)) * p.W + (ok ? ox : 0)) * p.out_stride + cb;
#pragma unroll
            for (int rh = 0; rh < 4; ++rh) {
                // kinds: [ph 0][0] = M0 + M1, [ph 0][1] = M1, [ph 1][0] = M2, [ph 1][1] = M3 of plane row i at ((nw 4 + i) 4 + kind) 256 + rg 64
                f32x4 in[1][3][3], bias4[1];
#pragma unroll
                for (int r2 = 0; r2 < 1; ++r2) {
                    const int rg = rh + r2;
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const f32x4 *x = xch + ((nw * 4 + pp + r) * 4) * 256 + rg * 64;
                        if (qq == 0) { in[r2][r][0] = x[0 * 256]; in[r2][r][1] = x[2 * 256]; }
                        else { in[r2][r][0] = x[1 * 256]; in[r2][r][1] = x[2 * 256]; in[r2][r][2] = x[3 * 256]; }
                    }
                    bias4[r2] = smem4[bias_q + ((cb + 8 * rg) >> 2)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r2 = 0; r2 < 1; ++r2) {
                    const int rg = rh + r2;
                    f32x4 R[3];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
                        R[r] = qq == 0 ? in[r2][r][0] + in[r2][r][1] : sub4(sub4(in[r2][r][0], in[r2][r][1], m1), in[r2][r][2], m1);
                    f32x4 y;
                    if (pp == 0) y = (R[0] + R[1]) + R[2];
                    else y = sub4(sub4(R[0], R[1], m1), R[2], m1);
                    y += bias4[r2];
                    if (p.lrelu) {
                        y[0] = lrelu01(y[0]); y[1] = lrelu01(y[1]); y[2] = lrelu01(y[2]); y[3] = lrelu01(y[3]);
                    }
                    if (ok && cb + 8 * rg < p.cout_store) *reinterpret_cast<f32x4 *>(orow + 8 * rg) = y;
                }
            }
        }
    };

    // Producer step c (PAR = c & 1 at compile time: the LDS addresses are registers plus immediates): commit raw(c + 2) from the
    // register set loaded two steps ago, fetch raw(c + 4) into it, move the cursor.  In the first step of an item, the previous
    // item's output (vector work: it runs when the consumers have issued that step's MFMAs and wait at the barrier).
    int cs = 0, cc = 0;          // the consumers' (item, chunk) of the current step
#define WS_PSTEP(PAR, PR, LEADIN)                                                                 \
    do {                                                                                          \
        WS_STAMP(d_a_);                                                                           \
        WS_COMMIT(PR, PAR);                                                                       \
        WS_STAMP(d_b_);                                                                           \
        WS_LOADP(PR);                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        WS_ADVANCE();                                                                             \
        if (!(LEADIN)) {                                                                          \
            if (cc == 0 && cs > 0 && !WS_DBG(8)) finish_item(cs - 1);                             \
            if (++cc == nchunk) { cc = 0; ++cs; }                                                 \
        }                                                                                         \
        WS_SYNC();                                                                                \
    } while (0)

    WS_ENTER_ITEM(0);
    WS_LOADP(prA);                     // raw(0)
    WS_ADVANCE();
    WS_LOADP(prB);                     // raw(1)
    WS_ADVANCE();
    WS_PSTEP(0, prA, true);            // step -2: commit raw(0), fetch raw(2)
    WS_PSTEP(1, prB, true);            // step -1: commit raw(1), fetch raw(3)
    const int S = n_items * nchunk;
    for (int c = 0; c < S; c += 2) {
        WS_PSTEP(0, prA, false);
        if (c + 1 < S) WS_PSTEP(1, prB, false);
    }
    if (!WS_DBG(8)) finish_item(n_items - 1);
#ifdef PIVLFN_STAMPS
    if (stamp_ && lane == 0) {
        unsigned long long *o = p.stamps + (size_t)blockIdx.x * 16 + 8;
        o[0] = d_a_; o[1] = d_b_; o[2] = d_bar_; o[3] = __builtin_amdgcn_s_memtime() - t_begin_;
    }
#endif
#undef WS_MAKE_RS
#undef WS_ENTER_ITEM
#undef WS_ADVANCE
#undef WS_LOAD2
#undef WS_LOADP
#undef WS_COMMIT
#undef WS_PSTEP
#undef WS_SYNC
#undef WS_STAMP
#undef WS_DBG
}

template <int NBW>
static int launch_ws(const ConvParamsW &p, int grid, hipStream_t st)
{
    const size_t lds = (size_t)(WS_XQ + NBW * 4096) * 16 + (size_t)p.cout_pad * 4;       // 146 KB (two channel blocks) / 82 KB + the bias: one workgroup per CU
    static LdsAttr attr;
    if (int rc = ensure_dyn_lds(attr, reinterpret_cast<const void *>(conv_wino_ws_kernel<NBW>), (int)lds)) return rc;
    hipLaunchKernelGGL((conv_wino_ws_kernel<NBW>), dim3((unsigned)grid), dim3(768), lds, st, p);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

// Work items (32 tiles x 32 NBW channels) a launch of the specialised kernel would have; 0 = the layer is not covered.
long conv_wino_ws_items(const ConvParamsW &p)
{
    const int nb = p.cout_pad / 32;
    if (p.nchunk < 2) return 0;
    // the interior-item load offsets span 10 rows of a source: 32-bit byte offsets
    for (int s = 0; s < p.nseg; ++s)
        if ((long)10 * p.W * p.seg[s].stride * 4 >= (1L << 31)) return 0;
    const int nbw = nb % 2 == 0 ? 2 : 1;
    return (long)p.B * cdiv(p.H, 8) * cdiv(p.W, 16) * (nb / nbw);
}

// Arguments are validated by launch_conv_w (conv_wino.hip), which dispatches here.
int launch_conv_w_ws(const ConvParamsW &p, hipStream_t st)
{
    const long items = conv_wino_ws_items(p);
    PIV_REQUIRE(items > 0 && items < (1L << 31), "conv_wino_ws: %ld work items", items);
    const int cus = device_cus() / 8 * 8;
    const int grid = (int)std::min<long>(std::max(cus, 8), (items + 7) / 8 * 8);
    return (p.cout_pad / 32) % 2 == 0 ? launch_ws<2>(p, grid, st) : launch_ws<1>(p, grid, st);
}

}  // namespace pivlfn
