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
//     never touch LDS, and LDS holds only the raw 8-channel input patch (double-buffered, one barrier per K chunk);
//   * epilogue: the column half of A^T M A in registers, the row half across the four waves through LDS (which is idle by
//     then), bias / LeakyReLU, 16-byte NHWC stores.
// Summation order per output value: chunks ascending, k = {j, 4+j} inside a chunk, then planes (fixed order): independent
// of MB, of the grid and of the batch, so the tile shape may be chosen from the launch size.
#include <algorithm>
#include <vector>
#include "common.h"

namespace pivlfn {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int WPW = 18;      // patch width in pixels: 8 tiles x 2 + 2
constexpr int WPIX = 8;      // floats per staged pixel = one K chunk
constexpr unsigned WOOB = 0x80000000u;

template <int MB>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(const ConvParamsW p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TY = 4 * MB;                  // tile rows of the workgroup
    constexpr int PH = 2 * TY + 2;
    constexpr int NSLOT = PH * WPW * 2;         // 16-byte slots of one chunk's patch
    constexpr int PS = (NSLOT + 255) / 256;
    constexpr int PBUF = (PH * WPW + 1) * WPIX; // floats per patch buffer (+ one spare record)

    const int NB = p.cout_pad >> 5;
    const int tiles_x = (p.W + 15) >> 4, tiles_y = (p.H + 2 * TY - 1) / (2 * TY);
    int t = xcd_remap(blockIdx.x, gridDim.x);   // the N blocks of a spatial tile run back to back on one XCD: its patch is fetched once into that L2
    const int nb = t % NB;
    t /= NB;
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
        ppix[s] = in ? (unsigned)(iy * p.W + ix) : WOOB;
        plds[s] = (idx < NSLOT ? pix : PH * WPW) * WPIX + q4;
    }
    // plane row i = wave: (B^T d)[i][.] = d[ra][.] + sb * d[rb][.]
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sb = wave == 1 ? 1.f : -1.f;
    int abase[MB], bbase[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int tyl = mb * 4 + (n >> 3), txl = n & 7;
        abase[mb] = ((2 * tyl + ra) * WPW + 2 * txl) * WPIX + 4 * g;
        bbase[mb] = ((2 * tyl + rb) * WPW + 2 * txl) * WPIX + 4 * g;
    }

    f32x16 acc[4][MB];
#pragma unroll
    for (int jp = 0; jp < 4; ++jp)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[jp][mb][r] = 0.f;

    // weights of (chunk, nb, wave): 4 planes x 64 lanes x 4 floats, contiguous
    const float *wlane = p.wpk + ((size_t)nb * 4 + wave) * 1024 + lane * 4;
    const size_t wchunk = (size_t)NB * 4 * 1024;

    int seg = 0, c0 = 0, lchunk = 0;
    int scl = p.seg[0].cload, sst4 = p.seg[0].stride * 4;
    const size_t img_px = (size_t)p.H * p.W;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.seg[0].ptr + (size_t)b * img_px * p.seg[0].stride), 0,
                                                                    (unsigned)(((img_px - 1) * p.seg[0].stride + p.seg[0].cload) * 4), 0x00020000);
    f32x4 pr[PS], wn[4], wc[4];

#define WINO_LOAD()                                                                               \
    do {                                                                                          \
        const unsigned coff_ = (unsigned)(c0 + q4) * 4u;                                          \
        const bool qok_ = c0 + q4 < scl;                                                          \
        _Pragma("unroll") for (int s = 0; s < PS; ++s) {                                          \
            const unsigned o_ = (ppix[s] != WOOB && qok_) ? ppix[s] * (unsigned)sst4 + coff_ : WOOB; \
            pr[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)o_, 0, 0)); \
        }                                                                                         \
        const float *w_ = wlane + (size_t)lchunk * wchunk;                                        \
        _Pragma("unroll") for (int jp = 0; jp < 4; ++jp) wn[jp] = *reinterpret_cast<const f32x4 *>(w_ + jp * 256); \
    } while (0)
#define WINO_COMMIT(BUF)                                                                          \
    do {                                                                                          \
        _Pragma("unroll") for (int s = 0; s < PS; ++s) *reinterpret_cast<f32x4 *>(smem + (BUF)*PBUF + plds[s]) = pr[s]; \
    } while (0)

    WINO_LOAD();
    WINO_COMMIT(0);
    __syncthreads();
    for (int chunk = 0; chunk < p.nchunk; ++chunk) {
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) wc[jp] = wn[jp];
        // the next chunk's patch and weights go in flight before this chunk's matrix work; past the end the last chunk is
        // fetched again (into the buffer nobody reads any more), which keeps the body free of branches around loads
        if (lchunk + 1 < p.nchunk) {
            ++lchunk;
            c0 += 8;
            if (c0 >= scl) {
                ++seg;
                c0 = 0;
                scl = p.seg[seg].cload; sst4 = p.seg[seg].stride * 4;
                rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.seg[seg].ptr + (size_t)b * img_px * p.seg[seg].stride), 0,
                                                       (unsigned)(((img_px - 1) * p.seg[seg].stride + p.seg[seg].cload) * 4), 0x00020000);
            }
        }
        WINO_LOAD();
        __builtin_amdgcn_sched_barrier(0);      // the loads stay in front of the matrix work (the scheduler would sink them to their first use)
        const float *buf = smem + (chunk & 1) * PBUF;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            f32x4 tt[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 da = *reinterpret_cast<const f32x4 *>(buf + abase[mb] + c * WPIX);
                const f32x4 db = *reinterpret_cast<const f32x4 *>(buf + bbase[mb] + c * WPIX);
                tt[c] = da + sb * db;
            }
            f32x4 v[4];
            v[0] = tt[0] - tt[2];
            v[1] = tt[1] + tt[2];
            v[2] = tt[2] - tt[1];
            v[3] = tt[1] - tt[3];
#pragma unroll
            for (int jp = 0; jp < 4; ++jp)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[jp][mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[jp][j], v[jp][j], acc[jp][mb], 0, 0, 0);
        }
        WINO_COMMIT((chunk + 1) & 1);
        __syncthreads();
    }
#undef WINO_LOAD
#undef WINO_COMMIT

    // ---- output transform.  acc[jp][mb][4 rg + e] = M[(wave, jp)][cout 8 rg + 4 g + e][tile n of block mb]
    // column half (in registers): R[0] = M0 + M1 + M2, R[1] = M1 - M2 - M3; row half across waves: Y[0] = R_0 + R_1 + R_2, Y[1] = R_1 - R_2 - R_3
    f32x4 *xch = reinterpret_cast<f32x4 *>(smem);       // [mb][wave][q][rg][lane]
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            f32x4 m[4];
#pragma unroll
            for (int jp = 0; jp < 4; ++jp)
                m[jp] = f32x4{acc[jp][mb][4 * rg + 0], acc[jp][mb][4 * rg + 1], acc[jp][mb][4 * rg + 2], acc[jp][mb][4 * rg + 3]};
            xch[(((mb * 4 + wave) * 2 + 0) * 4 + rg) * 64 + lane] = (m[0] + m[1]) + m[2];
            xch[(((mb * 4 + wave) * 2 + 1) * 4 + rg) * 64 + lane] = (m[1] - m[2]) - m[3];
        }
    __syncthreads();
    const int pp = wave >> 1, qq = wave & 1;      // this wave finishes output pixel (pp, qq) of every tile
    f32x4 bias4[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) bias4[rg] = *reinterpret_cast<const f32x4 *>(p.bias + nb * 32 + 8 * rg + 4 * g);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int oy = y0 + 2 * (mb * 4 + (n >> 3)) + pp, ox = x0 + 2 * (n & 7) + qq;
        const bool ok = oy < p.H && ox < p.W;
        float *orow = p.out + (size_t)((b * p.H + (ok ? oy : 0)) * p.W + (ok ? ox : 0)) * p.out_stride + nb * 32 + 4 * g;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const f32x4 *x = xch + ((mb * 4 * 2 + qq) * 4 + rg) * 64 + lane;     // wave i at x[i * 2 * 4 * 64]
            f32x4 y;
            if (pp == 0) y = (x[0] + x[1 * 512]) + x[2 * 512];
            else y = (x[1 * 512] - x[2 * 512]) - x[3 * 512];
            y += bias4[rg];
            if (p.lrelu) {
                y[0] = lrelu01(y[0]); y[1] = lrelu01(y[1]); y[2] = lrelu01(y[2]); y[3] = lrelu01(y[3]);
            }
            if (ok && nb * 32 + 8 * rg + 4 * g < p.cout_store) *reinterpret_cast<f32x4 *>(orow + 8 * rg) = y;
        }
    }
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

template <int MB>
static int launch_w(const ConvParamsW &p, hipStream_t st)
{
    constexpr int PH = 8 * MB + 2;
    const size_t lds = std::max<size_t>((size_t)MB * 32768, (size_t)2 * (PH * WPW + 1) * WPIX * sizeof(float));
    static LdsAttr attr;
    if (int rc = ensure_dyn_lds(attr, reinterpret_cast<const void *>(conv_wino_kernel<MB>), (int)lds)) return rc;
    const long blocks = (long)p.B * cdiv(p.H, 8 * MB) * cdiv(p.W, 16) * (p.cout_pad / 32);
    PIV_REQUIRE(blocks < (1L << 31), "conv_wino: grid too large");
    hipLaunchKernelGGL((conv_wino_kernel<MB>), dim3((unsigned)blocks), dim3(256), lds, st, p);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

int launch_conv_w(const ConvParamsW &p, hipStream_t st)
{
    PIV_REQUIRE(p.nseg >= 1 && p.nseg <= 3 && p.wpk && p.bias && p.out, "conv_wino: bad arguments");
    PIV_REQUIRE(p.cout_pad % 32 == 0 && p.cout_store <= p.cout_pad && p.cout_store % 4 == 0 && p.out_stride % 4 == 0,
                "conv_wino: cout_pad=%d cout_store=%d out_stride=%d", p.cout_pad, p.cout_store, p.out_stride);
    PIV_REQUIRE(p.B > 0 && p.H > 0 && p.W > 0 && (long)p.B * p.H * p.W < (1L << 31), "conv_wino: bad shape");
    for (int s = 0; s < p.nseg; ++s)      // 32-bit byte offsets inside one image of a source
        PIV_REQUIRE((long)p.H * p.W * p.seg[s].stride * 4 < (1L << 31), "conv_wino: image of source %d exceeds 2 GiB", s);
    int nchunk = 0;
    for (int s = 0; s < p.nseg; ++s) {
        PIV_REQUIRE(p.seg[s].cload % 4 == 0 && p.seg[s].stride % 4 == 0 && p.seg[s].ptr, "conv_wino: segment %d misaligned", s);
        nchunk += (p.seg[s].cload + 7) / 8;
    }
    PIV_REQUIRE(nchunk == p.nchunk, "conv_wino: segments hold %d chunks, weights were packed for %d", nchunk, p.nchunk);
    // 16-row blocks while they still give every CU two workgroups; the results do not depend on the choice (header)
    const long blocks2 = (long)p.B * cdiv(p.H, 16) * cdiv(p.W, 16) * (p.cout_pad / 32);
    return blocks2 >= 512 ? launch_w<2>(p, st) : launch_w<1>(p, st);
}

}  // namespace pivlfn
