// 3x3 stride-1 convolution by Winograd's minimal filtering F(2x2, 3x3) on the gfx950 fp32 matrix cores.
//
// Covers the dense 3x3 / stride 1 / pad 1 torch.nn.Conv2d layers of the LiteFlowNet path -- conv_M, conv_S, conv_R and the
// second layers of NetC (/root/reference/src/models.py:77-101, 154-160, 197-204, 236-250): 95 % of the network's multiplies.
// fp32 operands, fp32 products, fp32 accumulation on v_mfma_f32_32x32x2_f32 (exact fma chains); what changes against the direct
// kernel of conv_mfma.hip is the algorithm, not the arithmetic: 16 multiplies per 2x2 output tile and input channel instead
// of 36 (the same minimal-filtering algorithm cuDNN / MIOpen pick for fp32 3x3 layers):
//     Y = A^T [ (G g G^T) . (B^T d B) ] A ,   d = 4x4 input patch, g = 3x3 filter, Y = 2x2 outputs,
//     B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1].
// U = G g G^T is formed once at load time in float64 and rounded to fp32; B^T d B and A^T M A are fp32 additions.
//
// Mapping (one workgroup = 4 waves = an 8 x 4*MB block of 2x2 tiles x 32 output channels, all 16 frequency planes):
//   * wave i owns plane row i (planes (i,0..3)): its four planes need only two of the patch's four rows, so each wave reads
//     8 (not 16) pixels per tile from the staged patch and no wave repeats another's transform arithmetic;
//   * per plane a 32 x 32 x K matrix product  D[cout][tile] += U[cout][cin] V[cin][tile]  (A = weights, B = transformed
//     activations): the B fragment comes straight out of the lane's transform registers, the A fragment straight from
//     global memory (packed at load time in fragment order: one contiguous 1 KB per wave-load, L2-resident) -- the weights
//     never touch LDS, and LDS holds only the raw 8-channel input patch (double-buffered, one barrier per K chunk;
//     stored de-interleaved so the stride-2 operand reads are conflict-free, see WPIXQ below);
//   * epilogue: the column half of A^T M A in registers, the row half across the four waves through LDS (which is idle by
//     then), bias / LeakyReLU, 16-byte NHWC stores.
// Summation order per output value: chunks ascending, k = {j, 4+j} inside a chunk, then planes (fixed order): independent
// of MB, of the grid and of the batch, so the tile shape may be chosen from the launch size.
#include <algorithm>
#include <vector>
#include "common.h"
#include "wino_common.h"

namespace pivlfn {

// MB = 32-tile blocks per wave (the workgroup covers 8 x 4 MB tiles = 16 x 8 MB pixels), NBW = 32-channel output blocks per wave
// (the workgroup covers 32 NBW channels).  A transformed operand V feeds NBW MFMAs and a weight fragment MB MFMAs; on this chip
// the fp32 MFMA and the fp32 vector instructions share the SIMD's fp32 datapath (tools/micro/mfma_valu_mix.hip: every vector
// instruction beside v_mfma_f32_32x32x2_f32 costs the matrix work its 4 issue cycles), so the transform's two additions per V
// are paid in matrix-pipe time and NBW is what amortises them.  4 MB NBW accumulators of 16 registers: up to 8 (128 registers)
// two workgroups share a CU, 16 take the whole register file of one.
template <int MB, int NBW>
__global__ __launch_bounds__(256, (MB * NBW <= 2) ? 2 : 1) void conv_wino_kernel(const ConvParamsW p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TY = 4 * MB;                  // tile rows of the workgroup
    constexpr int PH = 2 * TY + 2;
    constexpr int NSLOT = PH * WPW * 2;         // 16-byte slots of one chunk's patch
    constexpr int PS = (NSLOT + 255) / 256;
    constexpr int PBUF = PH * WROWQ + WPIXQ;     // quads per patch buffer (+ one spare record)

    const int NB = p.cout_pad >> 5;             // 32-channel blocks of the layer
    const int NG = NB / NBW;                    // channel groups = workgroups per spatial tile
    const int tiles_x = (p.W + 15) >> 4, tiles_y = (p.H + 2 * TY - 1) / (2 * TY);
    int t = xcd_remap(blockIdx.x, gridDim.x);   // the channel groups of a spatial tile run back to back on one XCD: its patch is fetched once into that L2
    const int nb0 = (t % NG) * NBW;
    t /= NG;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int x0 = tx * 16, y0 = ty * 2 * TY;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, g = lane >> 5;

    // staging slots of this thread: slot s covers (pixel, quad) = (idx >> 1, idx & 1), idx = tid + 256 s.  Loads are buffer loads
    // through a per-image descriptor: a slot outside the image (the zero padding), past the patch, or past the source's channels
    // gets an out-of-range offset and returns zeros -- no divergent branch around a load, so the loop body stays straight-line
    // and the compiler's waits stay counted.  Slots past the patch land in a spare LDS record nobody reads.
    // The descriptors start at the first image row of this workgroup's patch (64-bit scalar arithmetic) and the 32-bit per-lane
    // offsets only span the patch rows: no limit on the size of one image (round 3 put the base at the image start and refused
    // images of 2 GiB per source -- a 2048 x 2048 pair's 128-channel level-1 tensors).
    const int row0 = max(y0 - 1, 0);
    unsigned ppix[PS];
    int plds[PS];
    const int q4 = (tid & 1) * 4;
#pragma unroll
    for (int s = 0; s < PS; ++s) {
        const int idx = tid + 256 * s;
        const int pix = idx >> 1;
        const int py = pix / WPW, px = pix - py * WPW;
        const int iy = y0 - 1 + py, ix = x0 - 1 + px;
        const bool in = idx < NSLOT && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        ppix[s] = in ? (unsigned)((iy - row0) * p.W + ix) : WOOB;       // relative to the descriptor's first row (below)
        plds[s] = (idx < NSLOT ? ((py >> 1) + (py & 1) * (PH / 2)) * WROWQ + ((px >> 1) + (px & 1) * 9) * WPIXQ : PH * WROWQ) + (tid & 1);
    }
    // plane row i = wave: (B^T d)[i][.] = d[ra][.] + sb * d[rb][.]
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sb = wave == 1 ? 1.f : -1.f;
    float m1 = -1.f;
    asm("" : "+v"(m1));
    int abase[MB], bbase[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int tyl = mb * 4 + (n >> 3), txl = n & 7;
        abase[mb] = (tyl + (ra >> 1) + (ra & 1) * (PH / 2)) * WROWQ + txl * WPIXQ + g;      // patch pixel (2 tyl + ra, 2 txl), in quads
        bbase[mb] = (tyl + (rb >> 1) + (rb & 1) * (PH / 2)) * WROWQ + txl * WPIXQ + g;
    }

    // (no zero initialisation: the first chunk's MFMAs take the constant 0 as their C operand -- 64 or 128 v_mov fewer in a prologue
    //  whose vector instructions queue behind the co-resident workgroup's MFMAs)
    f32x16 acc[4][MB][NBW];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // weights of (chunk, nb, wave): 4 planes x 64 lanes x 4 floats, contiguous; the wave's NBW blocks are 4096 floats apart
    const float *wbase = p.wpk + ((size_t)nb0 * 4 + wave) * 1024;
    const size_t wchunk = (size_t)NB * 4 * 1024;

    // per-source descriptors and strides, built once (scalar registers); `seg`, `c0`, `lchunk` = the chunk whose patch is loaded next
    const size_t img_px = (size_t)p.H * p.W;
    __amdgpu_buffer_rsrc_t rsv[3];
    int sclv[3], sst4v[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int ss = s < p.nseg ? s : 0;
        sclv[s] = p.seg[ss].cload;
        sst4v[s] = p.seg[ss].stride * 4;
        const size_t left = ((size_t)(p.H - row0) * p.W - 1) * p.seg[ss].stride + p.seg[ss].cload;       // floats from the base to the end of the image
        rsv[s] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.seg[ss].ptr + ((size_t)b * img_px + (size_t)row0 * p.W) * p.seg[ss].stride), 0,
                                                   (unsigned)min(left * 4, (size_t)0x7fffffff), 0x00020000);
    }
    int seg = 0, c0 = 0, lchunk = 0;
    unsigned pvo[PS];             // byte offset of the slot's pixel record (+ its quad) inside the current source, WOOB = none
#pragma unroll
    for (int s = 0; s < PS; ++s) pvo[s] = ppix[s] != WOOB ? ppix[s] * (unsigned)sst4v[0] + (unsigned)q4 * 4u : WOOB;
    f32x4 pr[PS], wA[NBW][4], wB[NBW][4], vA[4], vB[4];

// patch of chunk `lchunk` -> pr (voffset = the slot's record, soffset = the chunk's channel offset), then advance to the next chunk;
// past the end the last chunk is fetched again (into a buffer nobody reads any more): the loop body has no branch around a load
#define WINO_LOADP(PR)                                                                            \
    do {                                                                                          \
        const int scl_ = seg == 0 ? sclv[0] : (seg == 1 ? sclv[1] : sclv[2]);                     \
        const __amdgpu_buffer_rsrc_t rs_ = seg == 0 ? rsv[0] : (seg == 1 ? rsv[1] : rsv[2]);      \
        const bool qok_ = c0 + q4 < scl_;                                                         \
        _Pragma("unroll") for (int s = 0; s < PS; ++s)                                            \
            PR[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_, (int)(qok_ ? pvo[s] : WOOB), c0 * 4, 0)); \
        if (lchunk + 1 < p.nchunk) {                                                              \
            ++lchunk;                                                                             \
            c0 += 8;                                                                              \
            if (c0 >= scl_) {                                                                     \
                ++seg;                                                                            \
                c0 = 0;                                                                           \
                const int sst4_ = seg == 1 ? sst4v[1] : sst4v[2];                                 \
                _Pragma("unroll") for (int s = 0; s < PS; ++s)                                    \
                    pvo[s] = ppix[s] != WOOB ? ppix[s] * (unsigned)sst4_ + (unsigned)q4 * 4u : WOOB; \
            }                                                                                     \
        }                                                                                         \
    } while (0)
#define WINO_LOADW(WN, CH)                                                                        \
    do {                                                                                          \
        const float *w_ = wbase + (size_t)(CH)*wchunk;                                            \
        _Pragma("unroll") for (int nw = 0; nw < NBW; ++nw)                                        \
            _Pragma("unroll") for (int jp = 0; jp < 4; ++jp)                                      \
                WN[nw][jp] = *reinterpret_cast<const f32x4 *>(w_ + nw * 4096 + jp * 256 + lane * 4); \
    } while (0)
#define WINO_COMMIT(PR, BOFF)                                                                     \
    do {                                                                                          \
        _Pragma("unroll") for (int s = 0; s < PS; ++s) smem4[(BOFF) + plds[s]] = PR[s];          \
    } while (0)
// the lane's 8 operand quads of block MB_ (rows ra, rb x columns 0..3 of its tile's patch) from the patch image at quad offset BOFF
#define WINO_READ(RAW, BOFF, MB_)                                                                 \
    do {                                                                                          \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                           \
            RAW[c] = smem4[(BOFF) + abase[MB_] + ((c >> 1) + (c & 1) * 9) * WPIXQ];               \
            RAW[4 + c] = smem4[(BOFF) + bbase[MB_] + ((c >> 1) + (c & 1) * 9) * WPIXQ];           \
        }                                                                                         \
    } while (0)
// V = (B^T d B)[wave][0..3] as packed fp32 instructions on channel pairs: the row combination with the wave's sign and the three
// column differences as a + m * b (v_pk_fma_f32, m = +-1: exact products), the column sum as v_pk_add_f32.
#define WINO_XFORM(V, RAW)                                                                        \
    do {                                                                                          \
        f32x4 tt_[4];                                                                             \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) tt_[c] = sub4(RAW[c], RAW[4 + c], sb);      \
        V[0] = sub4(tt_[0], tt_[2], m1);                                                          \
        V[1] = tt_[1] + tt_[2];                                                                   \
        V[2] = sub4(tt_[2], tt_[1], m1);                                                          \
        V[3] = sub4(tt_[1], tt_[3], m1);                                                          \
    } while (0)
#define WINO_MFMA(V, WC, MB_, Z)                                                                  \
    do {                                                                                          \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                             \
            _Pragma("unroll") for (int jp = 0; jp < 4; ++jp)                                      \
                _Pragma("unroll") for (int nw = 0; nw < NBW; ++nw)                                \
                    acc[jp][MB_][nw] = __builtin_amdgcn_mfma_f32_32x32x2f32(WC[nw][jp][j], V[jp][j], ((Z) && j == 0) ? zero16 : acc[jp][MB_][nw], 0, 0, 0); \
    } while (0)
// One chunk: multiply chunk CH with (WC, VC); put chunk CH + 1's weights (-> WN) and chunk CH + 2's patch (-> pr -> LDS) in flight;
// compute the first block's transform of chunk CH + 1 (-> VN) under the matrix work.
#define WINO_STEP(WC, VC, WN, VN, CH, Z)                                                           \
    do {                                                                                          \
        WINO_LOADW(WN, (CH) + 1 < p.nchunk ? (CH) + 1 : (CH));                                    \
        WINO_LOADP(pr);                                                                           \
        __builtin_amdgcn_sched_barrier(0);      /* the loads stay in front of the matrix work (the scheduler would sink them to their first use) */ \
        WINO_MFMA(VC, WC, 0, Z);                                                                  \
        if (MB == 2) {                                                                            \
            f32x4 v1_[4];                                                                         \
            WINO_READ(raw, bo0, MB - 1);                                                          \
            WINO_XFORM(v1_, raw);                                                                 \
            WINO_MFMA(v1_, WC, MB - 1, Z);                                                        \
        }                                                                                         \
        WINO_READ(raw, bo1, 0);                                                                   \
        WINO_XFORM(VN, raw);                                                                      \
        WINO_COMMIT(pr, bo2);                                                                     \
        __syncthreads();                                                                          \
        const int t_ = bo0; bo0 = bo1; bo1 = bo2; bo2 = t_;                                       \
    } while (0)

// NBW = 2 with ONE weight set: the two channel blocks are multiplied one after the other (16 MFMAs each) and a block's registers
// receive the next chunk's fragments as soon as its MFMAs are issued -- half a step of latency cover instead of a whole one, but
// 32 registers less than two sets, which is what lets 8 accumulators + two channel blocks fit two workgroups per CU without spills.
#define WINO_MFMA_NW_J(V, NW, MB_, Z, J0, J1)                                                     \
    do {                                                                                          \
        _Pragma("unroll") for (int j = (J0); j < (J1); ++j)                                       \
            _Pragma("unroll") for (int jp = 0; jp < 4; ++jp)                                      \
                acc[jp][MB_][NW] = __builtin_amdgcn_mfma_f32_32x32x2f32(wA[NW][jp][j], V[jp][j], ((Z) && j == 0) ? zero16 : acc[jp][MB_][NW], 0, 0, 0); \
    } while (0)
#define WINO_MFMA_NW(V, NW, MB_, Z) WINO_MFMA_NW_J(V, NW, MB_, Z, 0, 4)
#define WINO_LOADW_NW(NW, CH)                                                                     \
    do {                                                                                          \
        const float *w_ = wbase + (size_t)(CH)*wchunk + (NW)*4096 + lane * 4;                     \
        _Pragma("unroll") for (int jp = 0; jp < 4; ++jp) wA[NW][jp] = *reinterpret_cast<const f32x4 *>(w_ + jp * 256); \
    } while (0)
#ifdef PIVLFN_STAMPS
#define WSTAMP(ACC)                                                                               \
    do {                                                                                          \
        if (stamp_) {                                                                             \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                         \
            ACC += now_ - tk_;                                                                    \
            tk_ = now_;                                                                           \
        }                                                                                         \
    } while (0)
#else
#define WSTAMP(ACC) do { } while (0)
#endif
#define WINO_STEP2(VC, VN, CH, Z)                                                                  \
    do {                                                                                          \
        const int nx_ = (CH) + 1 < p.nchunk ? (CH) + 1 : (CH);                                    \
        /* The first four MFMAs go out right behind the barrier and the patch loads' address work (scalar selects of the source, two \
           compares) issues under them instead of in front of them: -1 ... -2.5 % per layer.  The fence pins them behind the barrier \
           (left free the compiler hoists three of them above the previous step's commit: no faster). */ \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        WINO_MFMA_NW_J(VC, 0, 0, Z, 0, 1);                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        WINO_LOADP(pr);                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        WINO_MFMA_NW_J(VC, 0, 0, Z, 1, 4);                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        WINO_LOADW_NW(0, nx_);                                                                    \
        WSTAMP(d_blk0_);                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        WINO_MFMA_NW_J(VC, 1, 0, Z, 0, 3);                                                        \
        WINO_READ(raw, bo1, 0);                                                                   \
        WINO_XFORM(VN, raw);                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        /* the commit (wait for the patch loads, two LDS writes) goes in front of the block's last four MFMAs, so that the matrix \
           pipe still has ~256 cycles of this wave's work queued while the wave walks into the barrier: -1 ... -1.7 % per layer */ \
        WINO_COMMIT(pr, bo2);                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        WINO_MFMA_NW_J(VC, 1, 0, Z, 3, 4);                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        WSTAMP(d_blk1_);                                                                          \
        WINO_LOADW_NW(1, nx_);                                                                    \
        WSTAMP(d_commit_);                                                                        \
        __syncthreads();                                                                          \
        WSTAMP(d_bar_);                                                                           \
        const int t_ = bo0; bo0 = bo1; bo1 = bo2; bo2 = t_;                                       \
    } while (0)

    // Software pipeline over three patch buffers.  The step of chunk c multiplies chunk c; the patch of chunk c + 2 is in flight
    // (global -> registers) during it and committed to LDS at its end, so after the barrier that closes step c - 1 the image of
    // chunk c + 1 is already readable: the first block's transform of chunk c + 1 is computed inside step c, under the matrix
    // work of chunk c, and the first MFMA of a step issues right behind the barrier.  The loop is unrolled twice with the two
    // register sets (weights, first operand) swapping roles: no copies.
#ifdef PIVLFN_STAMPS
    const bool stamp_ = p.stamps != nullptr && wave == 0 && blockIdx.x < 8192;
    unsigned long long tk_ = 0, t_begin_ = 0, d_blk0_ = 0, d_blk1_ = 0, d_commit_ = 0, d_bar_ = 0, d_pro_ = 0;
    if (stamp_) t_begin_ = tk_ = __builtin_amdgcn_s_memtime();
#endif
    f32x4 *smem4 = reinterpret_cast<f32x4 *>(smem);
    int bo0 = 0, bo1 = PBUF, bo2 = 2 * PBUF;       // quad offsets of the buffers holding chunks c, c + 1, c + 2
    {   // prologue: the first two patches and the first weights all in flight together (one memory round trip, not two)
        f32x4 pr2[PS];
        WINO_LOADW(wA, 0);
        WINO_LOADP(pr);
        WINO_LOADP(pr2);
        WINO_COMMIT(pr, bo0);
        WINO_COMMIT(pr2, bo1);
    }
    __syncthreads();
    f32x4 raw[8];
    WINO_READ(raw, bo0, 0);
    WINO_XFORM(vA, raw);
    int chunk = 0;
    WSTAMP(d_pro_);
    // the first chunk is peeled: its MFMAs start the accumulators from the constant 0
    if constexpr (MB == 1 && NBW == 2) {
        WINO_STEP2(vA, vB, 0, true);
        for (chunk = 1; chunk + 1 < p.nchunk; chunk += 2) {
            WINO_STEP2(vB, vA, chunk, false);
            WINO_STEP2(vA, vB, chunk + 1, false);
        }
        if (chunk < p.nchunk) WINO_STEP2(vB, vA, chunk, false);
    } else {
        WINO_STEP(wA, vA, wB, vB, 0, true);
        for (chunk = 1; chunk + 1 < p.nchunk; chunk += 2) {
            WINO_STEP(wB, vB, wA, vA, chunk, false);
            WINO_STEP(wA, vA, wB, vB, chunk + 1, false);
        }
        if (chunk < p.nchunk) WINO_STEP(wB, vB, wA, vA, chunk, false);
    }
#undef WINO_LOADP
#undef WINO_LOADW
#undef WINO_COMMIT
#undef WINO_READ
#undef WINO_XFORM
#undef WINO_MFMA
#undef WINO_STEP
#undef WINO_STEP2
#undef WSTAMP
#undef WINO_MFMA_NW
#undef WINO_MFMA_NW_J
#undef WINO_LOADW_NW

#ifdef PIVLFN_STAMPS
    unsigned long long t_loop_end_ = 0;
    if (stamp_) t_loop_end_ = __builtin_amdgcn_s_memtime();
#endif
    // ---- output transform.  acc[jp][mb][nw][4 rg + e] = M[(wave, jp)][cout 32 (nb0 + nw) + 8 rg + 4 g + e][tile n of block mb]
    // column half (in registers): R[0] = M0 + M1 + M2, R[1] = M1 - M2 - M3; row half across waves: Y[0] = R_0 + R_1 + R_2, Y[1] = R_1 - R_2 - R_3
    f32x4 *xch = reinterpret_cast<f32x4 *>(smem);       // [mb][nw][wave][q][rg][lane]
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nw = 0; nw < NBW; ++nw)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                f32x4 m[4];
#pragma unroll
                for (int jp = 0; jp < 4; ++jp)
                    m[jp] = f32x4{acc[jp][mb][nw][4 * rg + 0], acc[jp][mb][nw][4 * rg + 1], acc[jp][mb][nw][4 * rg + 2], acc[jp][mb][nw][4 * rg + 3]};
                xch[((((mb * NBW + nw) * 4 + wave) * 2 + 0) * 4 + rg) * 64 + lane] = (m[0] + m[1]) + m[2];
                xch[((((mb * NBW + nw) * 4 + wave) * 2 + 1) * 4 + rg) * 64 + lane] = sub4(sub4(m[1], m[2], m1), m[3], m1);
            }
    __syncthreads();
    const int pp = wave >> 1, qq = wave & 1;      // this wave finishes output pixel (pp, qq) of every tile
#pragma unroll
    for (int nw = 0; nw < NBW; ++nw) {
        const int cb = (nb0 + nw) * 32 + 4 * g;
        f32x4 bias4[4];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) bias4[rg] = *reinterpret_cast<const f32x4 *>(p.bias + cb + 8 * rg);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int oy = y0 + 2 * (mb * 4 + (n >> 3)) + pp, ox = x0 + 2 * (n & 7) + qq;
            const bool ok = oy < p.H && ox < p.W;
            float *orow = p.out + (size_t)((b * p.H + (ok ? oy : 0)) * p.W + (ok ? ox : 0)) * p.out_stride + cb;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const f32x4 *x = xch + (((mb * NBW + nw) * 4 * 2 + qq) * 4 + rg) * 64 + lane;     // wave i at x[i * 2 * 4 * 64]
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
    }
#ifdef PIVLFN_STAMPS
    if (stamp_ && lane == 0) {
        const unsigned long long t_end_ = __builtin_amdgcn_s_memtime();
        unsigned long long *o = p.stamps + (size_t)blockIdx.x * 8;
        o[0] = d_pro_; o[1] = d_blk0_; o[2] = d_blk1_; o[3] = d_commit_; o[4] = d_bar_;
        o[5] = t_end_ - t_loop_end_;      // epilogue
        o[6] = t_end_ - t_begin_;
        o[7] = t_begin_;
    }
#endif
}

// OIHW [cout][cin][3][3] -> Winograd-domain weights in MFMA A-fragment order:
//   [chunk][n block][plane row i][plane col j][lane 64][4]: lane = (cout & 31) + 32 * k-half, element e multiplies staged
//   channel 8 * chunk_in_source + 4 * k-half + e of the chunk's source.  U = G g G^T in float64, rounded once to fp32.
void pack_conv_w(const float *w, int cout, int cin, const int *creal, const int *cload, const int *coff, int nseg,
                 std::vector<float> &pk, int *nchunk_out)
{
    const int cp = (cout + 31) / 32 * 32, NB = cp / 32;
    int nchunk = 0;
    for (int s = 0; s < nseg; ++s) nchunk += (cload[s] + 7) / 8;
    pk.assign((size_t)nchunk * NB * 16 * 256, 0.f);
    static const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    int chunk = 0, run = 0;
    for (int s = 0; s < nseg; ++s) {
        const int off = coff[s] >= 0 ? coff[s] : run;
        for (int c0 = 0; c0 < cload[s]; c0 += 8, ++chunk)
            for (int h = 0; h < 2; ++h)
                for (int e = 0; e < 4; ++e) {
                    const int c = c0 + 4 * h + e;
                    if (c >= creal[s]) continue;
                    for (int o = 0; o < cout; ++o) {
                        const float *gk = w + ((size_t)o * cin + off + c) * 9;
                        double tmp[4][3];
                        for (int i = 0; i < 4; ++i)
                            for (int x = 0; x < 3; ++x) tmp[i][x] = G[i][0] * gk[0 * 3 + x] + G[i][1] * gk[1 * 3 + x] + G[i][2] * gk[2 * 3 + x];
                        for (int i = 0; i < 4; ++i)
                            for (int j = 0; j < 4; ++j) {
                                const double u = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
                                const int nbk = o >> 5, ln = (o & 31) + 32 * h;
                                pk[(((((size_t)chunk * NB + nbk) * 4 + i) * 4 + j) * 64 + ln) * 4 + e] = (float)u;
                            }
                    }
                }
        run += creal[s];
    }
    *nchunk_out = nchunk;
}

bool conv_wino_supports(int KH, int KW, int S, int padY, int padX)
{
    return KH == 3 && KW == 3 && S == 1 && padY == 1 && padX == 1;
}

template <int MB, int NBW>
static int launch_w(const ConvParamsW &p, hipStream_t st)
{
    constexpr int PH = 8 * MB + 2;
    size_t lds = std::max<size_t>((size_t)MB * NBW * 32768, (size_t)3 * (PH * WROWQ + WPIXQ) * 16);
#ifdef PIVLFN_TOOLS
    if (PIV_KNOB(1) & 1048576) lds = 96 * 1024;      // measurement: one workgroup per CU (tools/bench_wino.py --masks 65536)
#endif
    static LdsAttr attr;
    if (int rc = ensure_dyn_lds(attr, reinterpret_cast<const void *>(conv_wino_kernel<MB, NBW>), (int)lds)) return rc;
    const long blocks = (long)p.B * cdiv(p.H, 8 * MB) * cdiv(p.W, 16) * (p.cout_pad / (32 * NBW));
    PIV_REQUIRE(blocks < (1L << 31), "conv_wino: grid too large");
    hipLaunchKernelGGL((conv_wino_kernel<MB, NBW>), dim3((unsigned)blocks), dim3(256), lds, st, p);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

int launch_conv_w(const ConvParamsW &p_in, hipStream_t st)
{
    ConvParamsW p = p_in;
    p.stamps = reinterpret_cast<unsigned long long *>(((unsigned long long)(unsigned)PIV_KNOB(6) << 32) | (unsigned)PIV_KNOB(5));   // tools only (0 in production)
    PIV_SET_DBG(p, PIV_KNOB(15));
    PIV_REQUIRE(p.nseg >= 1 && p.nseg <= 3 && p.wpk && p.bias && p.out, "conv_wino: bad arguments");
    PIV_REQUIRE(p.cout_pad % 32 == 0 && p.cout_store <= p.cout_pad && p.cout_store % 4 == 0 && p.out_stride % 4 == 0,
                "conv_wino: cout_pad=%d cout_store=%d out_stride=%d", p.cout_pad, p.cout_store, p.out_stride);
    PIV_REQUIRE(p.B > 0 && p.H > 0 && p.W > 0 && (long)p.B * p.H * p.W < (1L << 31), "conv_wino: bad shape");
    for (int s = 0; s < p.nseg; ++s)      // 32-bit byte offsets inside the rows of one patch (descriptors are rebased per workgroup)
        PIV_REQUIRE((long)20 * p.W * p.seg[s].stride * 4 < (1L << 31), "conv_wino: 20 rows of source %d exceed 2 GiB", s);
    int nchunk = 0;
    for (int s = 0; s < p.nseg; ++s) {
        PIV_REQUIRE(p.seg[s].cload % 4 == 0 && p.seg[s].stride % 4 == 0 && p.seg[s].ptr, "conv_wino: segment %d misaligned", s);
        nchunk += (p.seg[s].cload + 7) / 8;
    }
    PIV_REQUIRE(nchunk == p.nchunk, "conv_wino: segments hold %d chunks, weights were packed for %d", nchunk, p.nchunk);
    // Tile shape from the launch size; a value's summation order does not depend on it (header), so it may follow the batch.
    const int nb = p.cout_pad / 32;
    const long sp1 = (long)p.B * cdiv(p.H, 8) * cdiv(p.W, 16), sp2 = (long)p.B * cdiv(p.H, 16) * cdiv(p.W, 16);
    int mb = sp2 * nb >= 512 ? 2 : 1;
#ifdef PIVLFN_TOOLS
    // Round 5's persistent kernel with specialised waves (conv_wino_ws.hip): same values bit for bit (tests/test_gpu_wino.py), 4-8 % slower than this
    // kernel on the 128-channel layers and up to 24 % on 64 -> 32 / 32 -> 32 (profiles/r05_bench_wino_ws.log) -- compiled into the tools library only, for A/B runs (knob 14 = 31).
    if (PIV_KNOB(14) == 31 && conv_wino_ws_items(p) > 0) return launch_conv_w_ws(p, st);
#endif
#ifdef PIVLFN_TOOLS
    // A/B of the tile shapes (tools/bench_wino.py --masks): two or four channel blocks per wave halve / quarter the transform's
    // vector work per MFMA but run one workgroup per CU (16 accumulators) or spill (1 x 2): measured slower, see DESIGN.md
    const int force = PIV_KNOB(14);
    if (force == 22 && nb % 2 == 0) return launch_w<2, 2>(p, st);
    if (force == 14 && nb % 4 == 0) return launch_w<1, 4>(p, st);
    if (force == 21) return launch_w<2, 1>(p, st);
    if (force == 11) mb = 1;
#endif
    // two channel blocks per wave (half the transform work per MFMA, one weight set) wherever the layer has an even number of
    // 32-channel blocks: -4...-5 % against 16-row tiles with one block per wave on every such layer of levels 1 and 2
    if (nb % 2 == 0 && sp1 * (nb / 2) >= 256) return launch_w<1, 2>(p, st);
    if (mb == 2) return launch_w<2, 1>(p, st);
    return launch_w<1, 1>(p, st);
}

}  // namespace pivlfn
