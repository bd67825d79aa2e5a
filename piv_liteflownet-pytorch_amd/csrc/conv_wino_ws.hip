// Winograd F(2x2, 3x3) with specialised waves: persistent workgroups of eight waves, four that only multiply and four that do
// everything else.
//
// Same layers, same packed weights, same arithmetic and the same summation order per output value as conv_wino.hip (chunks
// ascending, k = {j, 4 + j} inside a chunk, the output transform's additions in the same order): the results are bit-identical to
// that kernel's, so which of the two runs a layer may follow the launch size.  What changes is who issues what.  In conv_wino.hip
// every wave loads, transforms, multiplies and stores, and on gfx950 a wave issues in order: each vector instruction in front of a
// v_mfma_f32_32x32x2_f32 is ~4 cycles in which that wave's SIMD starts no matrix work (1.66 vector instructions per MFMA, prologue
// and epilogue per workgroup: the matrix pipe is busy 75 % of a launch).  A vector instruction of ANOTHER wave of the SIMD costs the
// matrix pipe nothing (tools/micro/mfma_neighbour.hip: the MFMA wave keeps 64.0 cycles per instruction beside a vector wave).  So:
//
//   * One workgroup per CU (persistent, 512 threads): waves 0-3 -- one per SIMD -- are CONSUMERS, waves 4-7 -- again one per SIMD --
//     are PRODUCERS.  Consumer i owns plane row i (planes (i, 0..3)) of a 32-tile x 32 NBW-channel work item, as in conv_wino.hip.
//   * A consumer's K step is 16 NBW MFMAs per plane row and, behind each plane's MFMAs, the refill of exactly the registers that plane
//     just read: one ds_read_b128 of the next chunk's transformed operand V (written by the producers) and NBW buffer loads of the next
//     chunk's weight fragments (fragment order, L2-resident, scalar offsets: no address arithmetic).  No transform, no staging, no
//     vector instruction in the loop.
//   * The producers run ahead of the consumers through two rings in LDS: raw 8-channel patches (global -> registers -> LDS,
//     de-interleaved image of wino_common.h) and transformed operands V = B^T d B (16 planes x 64 lanes x 16 bytes per chunk).
//     Producer wave i forms plane row i for consumer i.  One s_barrier per K step is the only synchronisation: at the barrier that
//     ends step c, V(c + 2) is complete and V(c)'s slot is free.
//   * The work items of a workgroup (tile x channel group; its XCD's band, interleaved over the XCD's CUs) form ONE step sequence:
//     the producers' loads are four steps ahead and cross item boundaries, so a consumer goes from the last MFMA of an item to the
//     first of the next with only the column half of the output transform in between (it writes R = M A to LDS); the row half,
//     bias, LeakyReLU and the stores of item k are producer work during the first step of item k + 1.
#include <algorithm>
#include "common.h"
#include "wino_common.h"

namespace pivlfn {

namespace {

constexpr int WS_PH = 10;                              // patch rows of an 8 x 4 block of 2x2 tiles
constexpr int WS_NSLOT = WS_PH * WPW * 2;              // 16-byte staging slots of one chunk's patch (360)
constexpr int WS_PS = 2;                               // slots per producer thread
constexpr int WS_PBUF = WS_PH * WROWQ + WPIXQ;         // quads per raw patch buffer (+ one spare record)
constexpr int WS_VQ = 1152;                            // quad offset of the V ring (3 x 1024 quads)
constexpr int WS_XQ = WS_VQ + 3072;                    // quad offset of the output-transform exchange (NBW x 2048 quads)
static_assert(2 * WS_PBUF <= WS_VQ, "raw ring overlaps the V ring");

}  // namespace

template <int NBW>
__global__ __launch_bounds__(512) void conv_wino_ws_kernel(const ConvParamsW p)
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
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, g = lane >> 5;
    float m1 = -1.f;
    asm("" : "+v"(m1));
#ifdef PIVLFN_STAMPS
    // tools build: ablation mask p.dbg (1 no patch loads, 2 no weight refills, 4 no V refills, 8 no finish_item, 16 no column half,
    // 32 no transform) and per-workgroup stamps of wave 0 (consumer) and wave 4 (producer): ticks spent waiting at the step barriers
    const int dbg = p.dbg;
    const bool stamp_ = p.stamps != nullptr && (wave == 0 || wave == 4) && blockIdx.x < 4096;
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

    if (wave < 4) {
        // ------------------------------------------------------------------------------------------------ consumer, plane row `wave`
        if (!WS_DBG(64)) __builtin_amdgcn_s_setprio(3);
        f32x16 acc[4][NBW];
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x4 V[4], Wt[NBW][4];
        // weights of (chunk, block nb, plane row i): 4 planes x 64 lanes x 16 bytes, contiguous: byte offset ((chunk NB + nb) 4 + i) 4096
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.wpk), 0,
                                                                            (unsigned)((size_t)nchunk * NB * 16 * 1024), 0x00020000);
        const int wvoff = lane * 16;
        const int wstep = NB * 4 * 4096;              // bytes from one chunk to the next
        const int vlane = WS_VQ + wave * 256 + lane;  // quad index of the lane's operand of plane (wave, 0) in V slot 0

#define WS_REFILL(JP, VSLOT, SOFF)                                                                \
    do {                                                                                          \
        if (!WS_DBG(4)) V[JP] = smem4[vlane + (VSLOT)*1024 + (JP)*64];                            \
        if (!WS_DBG(2)) _Pragma("unroll") for (int nw = 0; nw < NBW; ++nw)                        \
            Wt[nw][JP] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wvoff + (JP)*1024, (SOFF) + nw * 16384, 0)); \
    } while (0)
// One K step: per plane 4 NBW MFMAs (k pairs ascending), then the refill of that plane's registers for the next step.
#define WS_CSTEP(Z, VSLOT, SOFF)                                                                  \
    do {                                                                                          \
        _Pragma("unroll") for (int jp = 0; jp < 4; ++jp) {                                        \
            __builtin_amdgcn_sched_barrier(0);                                                    \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                         \
                _Pragma("unroll") for (int nw = 0; nw < NBW; ++nw)                                \
                    acc[jp][nw] = __builtin_amdgcn_mfma_f32_32x32x2f32(Wt[nw][jp][j], V[jp][j], ((Z) && j == 0) ? zero16 : acc[jp][nw], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                    \
            WS_REFILL(jp, VSLOT, SOFF);                                                           \
        }                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    } while (0)

        auto item_woff = [&](int s) {
            const int t = t0 + s * tstep;
            return (((t % NG) * NBW) * 4 + wave) * 4096;
        };
        WS_SYNC();      // step -4: raw(0) committed
        WS_SYNC();      // step -3: V(0) written ...
        WS_SYNC();      // step -2: ... and complete
        int woff = item_woff(0);
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) WS_REFILL(jp, 0, woff);
        WS_SYNC();      // step -1
        int vslot = 1;        // V slot the next refills read (chunk index modulo 3)
#define WS_NEXT_VSLOT() vslot = vslot == 2 ? 0 : vslot + 1
        for (int s = 0; s < n_items; ++s) {
            const bool more = s + 1 < n_items;
            const int woff_next = more ? item_woff(s + 1) : 0;
            int soff = woff + wstep;
            WS_CSTEP(true, vslot, soff);
            WS_SYNC();
            WS_NEXT_VSLOT();
            for (int c = 1; c + 1 < nchunk; ++c) {
                soff += wstep;
                WS_CSTEP(false, vslot, soff);
                WS_SYNC();
                WS_NEXT_VSLOT();
            }
            // last chunk: the refills fetch the next item's first chunk (past the end: this chunk again, nobody uses it)
            soff = more ? woff_next : soff;
            WS_CSTEP(false, vslot, soff);
            // column half of the output transform: R[0] = M0 + M1 + M2, R[1] = M1 - M2 - M3 -> LDS [nw][plane row][q][rg][lane]
            f32x4 *xch = smem4 + WS_XQ;
            WS_STAMP(d_a_);
            if (!WS_DBG(16))
#pragma unroll
            for (int nw = 0; nw < NBW; ++nw)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    f32x4 m[4];
#pragma unroll
                    for (int jp = 0; jp < 4; ++jp)
                        m[jp] = f32x4{acc[jp][nw][4 * rg + 0], acc[jp][nw][4 * rg + 1], acc[jp][nw][4 * rg + 2], acc[jp][nw][4 * rg + 3]};
                    xch[(((nw * 4 + wave) * 2 + 0) * 4 + rg) * 64 + lane] = (m[0] + m[1]) + m[2];
                    xch[(((nw * 4 + wave) * 2 + 1) * 4 + rg) * 64 + lane] = sub4(sub4(m[1], m[2], m1), m[3], m1);
                }
            WS_STAMP(d_b_);
            WS_SYNC();
            WS_NEXT_VSLOT();
            woff = woff_next;
        }
#ifdef PIVLFN_STAMPS
        if (stamp_ && lane == 0) {
            unsigned long long *o = p.stamps + (size_t)blockIdx.x * 16;
            o[0] = d_a_; o[1] = d_b_; o[2] = d_bar_; o[3] = __builtin_amdgcn_s_memtime() - t_begin_; o[4] = t_begin_; o[5] = (unsigned long long)n_items;
        }
#endif
#undef WS_CSTEP
#undef WS_REFILL
#undef WS_NEXT_VSLOT
        return;
    }

    // ---------------------------------------------------------------------------------------------------- producer, plane row `pw`
    const int pw = wave - 4;
    const int ptid = tid - 256;
    if (WS_DBG(128)) __builtin_amdgcn_s_setprio(3);
    // staging slots of this thread: slot s covers (pixel, quad) = (idx >> 1, idx & 1), idx = ptid + 256 s (conv_wino.hip)
    int plds[WS_PS], ppy[WS_PS], ppx[WS_PS];
    const int q4 = (ptid & 1) * 4;
#pragma unroll
    for (int s = 0; s < WS_PS; ++s) {
        const int idx = ptid + 256 * s;
        const int pix = idx >> 1;
        const int py = pix / WPW, px = pix - py * WPW;
        ppy[s] = idx < WS_NSLOT ? py : -100000;
        ppx[s] = px;
        plds[s] = (idx < WS_NSLOT ? ((py >> 1) + (py & 1) * (WS_PH / 2)) * WROWQ + ((px >> 1) + (px & 1) * 9) * WPIXQ : WS_PH * WROWQ) + (ptid & 1);
    }
    // plane row i = pw: (B^T d)[i][.] = d[ra][.] + sb * d[rb][.]
    const int ra = pw == 0 ? 0 : (pw == 2 ? 2 : 1);
    const int rb = pw == 0 ? 2 : (pw == 1 ? 2 : (pw == 2 ? 1 : 3));
    const float sb = pw == 1 ? 1.f : -1.f;
    const int tyl = n >> 3, txl = n & 7;
    const int abase = (tyl + (ra >> 1) + (ra & 1) * (WS_PH / 2)) * WROWQ + txl * WPIXQ + g;      // patch pixel (2 tyl + ra, 2 txl), in quads
    const int bbase = (tyl + (rb >> 1) + (rb & 1) * (WS_PH / 2)) * WROWQ + txl * WPIXQ + g;
    const int vlane = WS_VQ + pw * 256 + lane;

    // ---- load cursor: the (item, chunk) whose raw patch is fetched next, and everything its load needs -- descriptor, scalar channel
    // offset and the per-slot byte offsets -- computed one step ahead, in the vector section of the step before
    const size_t img_px = (size_t)p.H * p.W;
    int l_item = 0, l_chunk = 0, l_seg = 0, l_c0 = 0;
    // per-source constants and descriptors as plain scalars (an array indexed by the run-time source goes to scratch memory and
    // takes the whole cursor into vector registers with it)
    const int scl0 = p.seg[0].cload, scl1 = p.seg[p.nseg > 1 ? 1 : 0].cload, scl2 = p.seg[p.nseg > 2 ? 2 : 0].cload;
    const int sst40 = p.seg[0].stride * 4, sst41 = p.seg[p.nseg > 1 ? 1 : 0].stride * 4, sst42 = p.seg[p.nseg > 2 ? 2 : 0].stride * 4;
    __amdgpu_buffer_rsrc_t rs0, rs1, rs2;
    unsigned ppix[WS_PS];
    auto decode = [&](int s, int &b, int &y0, int &x0, int &nb0) {
        int t = t0 + s * tstep;
        nb0 = (t % NG) * NBW;
        t /= NG;
        x0 = (t % tiles_x) * 16;
        t /= tiles_x;
        y0 = (t % tiles_y) * 8;
        b = t / tiles_y;
    };
    // the descriptor of source SS for the item whose patch starts at image row row0 of image b: it starts at that row (64-bit
    // scalar arithmetic), the 32-bit per-lane offsets only span the patch rows
#define WS_MAKE_RS(SS, B_, ROW0)                                                                  \
    __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.seg[SS].ptr + ((size_t)(B_)*img_px + (size_t)(ROW0)*p.W) * p.seg[SS].stride), 0, \
                                      (unsigned)min((((size_t)(p.H - (ROW0)) * p.W - 1) * p.seg[SS].stride + p.seg[SS].cload) * 4, (size_t)0x7fffffff), 0x00020000)
#define WS_ENTER_ITEM(S_)                                                                         \
    do {                                                                                          \
        int b_, y0_, x0_, nb0_;                                                                   \
        decode(S_, b_, y0_, x0_, nb0_);                                                           \
        const int row0_ = max(y0_ - 1, 0);                                                        \
        rs0 = WS_MAKE_RS(0, b_, row0_);                                                           \
        rs1 = WS_MAKE_RS(p.nseg > 1 ? 1 : 0, b_, row0_);                                          \
        rs2 = WS_MAKE_RS(p.nseg > 2 ? 2 : 0, b_, row0_);                                          \
        _Pragma("unroll") for (int k = 0; k < WS_PS; ++k) {                                       \
            const int iy = y0_ - 1 + ppy[k], ix = x0_ - 1 + ppx[k];                               \
            const bool in = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;                           \
            ppix[k] = in ? (unsigned)((iy - row0_) * p.W + ix) : WOOB;                            \
        }                                                                                         \
    } while (0)
    // what the NEXT load instruction takes (WS_PREPARE, from the cursor): descriptor, scalar offset, per-slot byte offsets
    __amdgpu_buffer_rsrc_t ld_rs;
    int ld_soff;
    unsigned ld_voff[WS_PS];
#define WS_PREPARE()                                                                              \
    do {                                                                                          \
        const int scl_ = l_seg == 0 ? scl0 : (l_seg == 1 ? scl1 : scl2);                          \
        const int sst4_ = l_seg == 0 ? sst40 : (l_seg == 1 ? sst41 : sst42);                      \
        ld_rs = l_seg == 0 ? rs0 : (l_seg == 1 ? rs1 : rs2);                                      \
        ld_soff = l_c0 * 4;                                                                       \
        const bool qok_ = l_c0 + q4 < scl_;                                                       \
        _Pragma("unroll") for (int k = 0; k < WS_PS; ++k)                                         \
            ld_voff[k] = (qok_ && ppix[k] != WOOB) ? ppix[k] * (unsigned)sst4_ + (unsigned)q4 * 4u : WOOB; \
    } while (0)
// next chunk, next source, next item; past the last item the last chunk is fetched again (into a buffer nobody reads)
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
        WS_PREPARE();                                                                             \
    } while (0)
    f32x4 prA[WS_PS] = {}, prB[WS_PS] = {};
// no vector instruction in these three: the producer issues them while its SIMD's consumer streams MFMAs
#define WS_LOADP(PR)                                                                              \
    do {                                                                                          \
        if (!WS_DBG(1)) _Pragma("unroll") for (int k = 0; k < WS_PS; ++k)                         \
            PR[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ld_rs, (int)ld_voff[k], ld_soff, 0)); \
    } while (0)
#define WS_COMMIT(PR, RSLOT)                                                                      \
    do {                                                                                          \
        _Pragma("unroll") for (int k = 0; k < WS_PS; ++k) smem4[(RSLOT)*WS_PBUF + plds[k]] = PR[k]; \
    } while (0)
#define WS_READRAW(RSLOT)                                                                         \
    do {                                                                                          \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                           \
            raw_[c] = smem4[(RSLOT)*WS_PBUF + abase + ((c >> 1) + (c & 1) * 9) * WPIXQ];          \
            raw_[4 + c] = smem4[(RSLOT)*WS_PBUF + bbase + ((c >> 1) + (c & 1) * 9) * WPIXQ];      \
        }                                                                                         \
    } while (0)
// V = (B^T d B)[pw][0..3] of the lane's (tile, k half) from raw_ -> V ring slot VSLOT (a run-time value: the first vector instruction)
#define WS_XFORM(VSLOT)                                                                           \
    do {                                                                                          \
        f32x4 tt_[4];                                                                             \
        f32x4 *vq_ = smem4 + vlane + (VSLOT)*1024;                                                \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) tt_[c] = sub4(raw_[c], raw_[4 + c], sb);    \
        vq_[0 * 64] = sub4(tt_[0], tt_[2], m1);                                                   \
        vq_[1 * 64] = tt_[1] + tt_[2];                                                            \
        vq_[2 * 64] = sub4(tt_[2], tt_[1], m1);                                                   \
        vq_[3 * 64] = sub4(tt_[1], tt_[3], m1);                                                   \
    } while (0)

    // row half of the output transform, bias, LeakyReLU and the stores of item s (its R planes are in the exchange area):
    // producer wave pw finishes output pixel (pp, qq) of every tile
    auto finish_item = [&](int s) {
        int b, y0, x0, nb0;
        decode(s, b, y0, x0, nb0);
        const int pp = pw >> 1, qq = pw & 1;
        const f32x4 *xch = smem4 + WS_XQ;
        const int oy = y0 + 2 * tyl + pp, ox = x0 + 2 * txl + qq;
        const bool ok = oy < p.H && ox < p.W;
#pragma unroll
        for (int nw = 0; nw < NBW; ++nw) {
            const int cb = (nb0 + nw) * 32 + 4 * g;
            f32x4 bias4[4];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) bias4[rg] = *reinterpret_cast<const f32x4 *>(p.bias + cb + 8 * rg);
            float *orow = p.out + (size_t)((b * p.H + (ok ? oy : 0)) * p.W + (ok ? ox : 0)) * p.out_stride + cb;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const f32x4 *x = xch + (((nw * 4) * 2 + qq) * 4 + rg) * 64 + lane;     // plane row i at x[i * 512]
                f32x4 y;
                if (pp == 0) y = (x[0] + x[1 * 512]) + x[2 * 512];
                else y = sub4(sub4(x[1 * 512], x[2 * 512], m1), x[3 * 512], m1);
                y += bias4[rg];
                if (p.lrelu) {
                    y[0] = lrelu01(y[0]); y[1] = lrelu01(y[1]); y[2] = lrelu01(y[2]); y[3] = lrelu01(y[3]);
                }
                if (ok && cb + 8 * rg < p.cout_store) *reinterpret_cast<f32x4 *>(orow + 8 * rg) = y;
            }
        }
    };

    // Producer step c, program order (PAR = c & 1 at compile time, so that every LDS address below is a register plus an immediate):
    //   commit raw(c + 4) from the register set loaded two steps ago; fetch raw(c + 6) into it; read raw(c + 3) from LDS -- no vector
    //   instruction so far: all of it issues beside the consumer's MFMAs -- then the vector section, which in practice runs when the
    //   consumer has issued its last MFMA of the step and waits at the barrier (tools/micro/ws_gap.hip): the transform of raw(c + 3)
    //   -> V(c + 3) (three-slot ring: the writes need not have landed before the NEXT barrier), the cursor and offsets of the next
    //   load and, in the first step of an item, the previous item's output.
    f32x4 raw_[8];
    int vs3 = 0;                 // V slot of chunk c + 3 = (c + 3) mod 3
    int cs = 0, cc = 0;          // the consumers' (item, chunk) of the current step
#define WS_PSTEP(PAR, PR, FIRST, XF)                                                              \
    do {                                                                                          \
        WS_STAMP(d_a_);                                                                           \
        WS_COMMIT(PR, PAR);                                                                       \
        WS_STAMP(d_b_);                                                                           \
        WS_LOADP(PR);                                                                             \
        if (XF) WS_READRAW((PAR) ^ 1);                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if ((XF) && !WS_DBG(32)) WS_XFORM(vs3);                                                   \
        if (XF) vs3 = vs3 == 2 ? 0 : vs3 + 1;                                                     \
        WS_ADVANCE();                                                                             \
        if (!(FIRST)) {                                                                           \
            if (cc == 0 && cs > 0 && !WS_DBG(8)) finish_item(cs - 1);                             \
            if (++cc == nchunk) { cc = 0; ++cs; }                                                 \
        }                                                                                         \
        WS_SYNC();                                                                                \
    } while (0)

    WS_ENTER_ITEM(0);
    WS_PREPARE();
    WS_LOADP(prA);                     // raw(0)
    WS_ADVANCE();
    WS_LOADP(prB);                     // raw(1)
    WS_ADVANCE();
    WS_PSTEP(0, prA, true, false);     // step -4: commit raw(0), fetch raw(2)
    WS_PSTEP(1, prB, true, true);      // step -3: commit raw(1), fetch raw(3), raw(0) -> V(0)
    WS_PSTEP(0, prA, true, true);      // step -2
    WS_PSTEP(1, prB, true, true);      // step -1
    const int S = n_items * nchunk;
    for (int c = 0; c < S; c += 2) {
        WS_PSTEP(0, prA, false, true);
        if (c + 1 < S) WS_PSTEP(1, prB, false, true);
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
#undef WS_PREPARE
#undef WS_ADVANCE
#undef WS_LOADP
#undef WS_COMMIT
#undef WS_READRAW
#undef WS_XFORM
#undef WS_PSTEP
#undef WS_SYNC
#undef WS_STAMP
#undef WS_DBG
}

template <int NBW>
static int launch_ws(const ConvParamsW &p, int grid, hipStream_t st)
{
    const size_t lds = (size_t)(WS_XQ + NBW * 2048) * 16;       // 133 KB (two channel blocks) / 100 KB
    static LdsAttr attr;
    if (int rc = ensure_dyn_lds(attr, reinterpret_cast<const void *>(conv_wino_ws_kernel<NBW>), (int)lds)) return rc;
    hipLaunchKernelGGL((conv_wino_ws_kernel<NBW>), dim3((unsigned)grid), dim3(512), lds, st, p);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

// Work items (32 tiles x 32 NBW channels) a launch of the specialised kernel would have; 0 = the layer is not covered.
long conv_wino_ws_items(const ConvParamsW &p)
{
    const int nb = p.cout_pad / 32;
    if (p.nchunk < 2) return 0;
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
