// Direct convolution with fp16 multiplicands and fp32 accumulation on the gfx950 matrix cores (BASELINE config #5,
// SURVEY build-plan item 9): the optional reduced-precision mode of the conv stacks.  NOT the headline fp32 path --
// conv_mfma.hip stays the default and the only kernel the fp32 parity tests and bench.py's headline line use.
//
// Same implicit-GEMM shape as conv_mfma.hip (workgroup = 4 waves = (4*MT rows x 32 px) x (32*NT channels), input patch and
// weight slab of one K chunk staged in LDS and reused by all KH*KW taps, next chunk prefetched global->registers under the
// MFMAs), with these differences:
//   * v_mfma_f32_32x32x16_f16: 16x the per-SIMD rate of the fp32 instruction, so the kernel is bound by the CU's load
//     path, not by the matrix pipe -- hence the 16-row tile (MT=4, the whole 512-register file, one workgroup per CU):
//     the weight slab a workgroup restages per chunk is amortised over twice the pixels;
//   * K chunk = 16 input channels = one MFMA k-step per tap; a lane's operand is one ds_read_b128 (8 halfs): lanes 0-31
//     take channels 0-7 of their pixel / output channel, lanes 32-63 channels 8-15;
//   * a source may be fp32 (converted to fp16 while it is staged: two 16-byte loads and two ds_write_b64 per pixel pair
//     item) or fp16 (one load, one ds_write_b128); the output is stored as fp32 or fp16.  Products of fp16 values are exact
//     in fp32 and the accumulation is fp32, so the only rounding this mode adds is that of the operands (2^-11 relative);
//   * LDS pixel pitch 48 bytes (16 halfs + 16 bytes of padding): 3 * pixel mod 16 visits all sixteen 16-byte slots of a
//     256-byte bank row, so the 16-lane groups of ds_read_b128 are conflict-free; weights are [tap][k-half][channel][8].
#include <vector>
#include "common.h"

namespace pivlfn {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int HPITCH = 24;      // halfs per staged pixel (16 channels + 8 of padding = 48 bytes)

template <int MT, int NT, int PM, int WM>
__global__ __launch_bounds__(256, (MT * NT * 16 + (2 * PM + WM) * 4 + 60 > 240) ? 1 : 2) void conv_f16_kernel(const ConvParamsH p)
{
    extern __shared__ __attribute__((aligned(16))) _Float16 hsmem[];
    constexpr int BN = NT * 32;
    constexpr int TH = 4 * MT;
    const int taps = p.KH * p.KW;
    const int PH = (TH - 1) * p.S + p.KH;
    const int PW = 31 * p.S + p.KW;
    const int npix = PH * PW;
    _Float16 *patch = hsmem;
    _Float16 *wts = hsmem + npix * HPITCH;
    float *lbias = reinterpret_cast<float *>(wts + taps * 2 * BN * 8);      // this workgroup's BN biases (see the epilogue)

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

    int abase[MT];      // half offset of this lane's pixel operand for tap (0,0)
#pragma unroll
    for (int m = 0; m < MT; ++m) abase[m] = ((wave * MT + m) * p.S * PW + row * p.S) * HPITCH + hh * 8;
    const int bbase = (hh * BN + row) * 8;

    // staging items: patch item i = pixel (tid>>1) + 128*i, channel half `sub`; weight item i = 16-byte unit tid + 256*i
    int poff[PM];       // pixel index inside the source (times its pixel stride later), -1 = outside the image, -2 = no item
#pragma unroll
    for (int i = 0; i < PM; ++i) {
        const int pix = (tid >> 1) + 128 * i;
        const int py = pix / PW, px = pix - py * PW;
        const int iy = iy0 + py, ix = ix0 + px;
        poff[i] = pix >= npix ? -2 : ((iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? (b * p.H + iy) * p.W + ix : -1);
    }
    const int nw = taps * 2 * BN;
    const f32x4 *wsrc = reinterpret_cast<const f32x4 *>(p.wpk) + n0;
    const size_t wchunk = (size_t)taps * 2 * p.cout_pad;

    f32x4 pr[2 * PM], wr[WM];
    int seg = 0, c0 = 0;
    const char *sp = reinterpret_cast<const char *>(p.seg[0].ptr);
    int scl = p.seg[0].cload, sst = p.seg[0].stride, sf16 = p.seg[0].f16;
    int ld_f16 = 0;     // dtype of the chunk sitting in pr[]

#define CH_LOAD(CH)                                                                               \
    do {                                                                                          \
        ld_f16 = sf16;                                                                            \
        if (PIV_DBG(p) & 2) break;                                                                     \
        if (sf16) {                                                                               \
            const bool ok_ = c0 + 8 * sub < scl;                                                  \
            _Pragma("unroll") for (int i = 0; i < PM; ++i) {                                      \
                f32x4 v = {0.f, 0.f, 0.f, 0.f};                                                   \
                if (poff[i] >= 0 && ok_)                                                          \
                    v = *reinterpret_cast<const f32x4 *>(sp + ((size_t)poff[i] * sst + c0 + 8 * sub) * 2); \
                pr[i] = v;                                                                        \
            }                                                                                     \
        } else {                                                                                  \
            const bool ok0_ = c0 + 4 * sub < scl, ok1_ = c0 + 8 + 4 * sub < scl;                  \
            _Pragma("unroll") for (int i = 0; i < PM; ++i) {                                      \
                f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};                       \
                if (poff[i] >= 0) {                                                               \
                    const char *q_ = sp + ((size_t)poff[i] * sst + c0 + 4 * sub) * 4;             \
                    if (ok0_) v0 = *reinterpret_cast<const f32x4 *>(q_);                          \
                    if (ok1_) v1 = *reinterpret_cast<const f32x4 *>(q_ + 32);                     \
                }                                                                                 \
                pr[2 * i] = v0;                                                                   \
                pr[2 * i + 1] = v1;                                                               \
            }                                                                                     \
        }                                                                                         \
        const f32x4 *wc_ = wsrc + (size_t)(CH)*wchunk;                                            \
        _Pragma("unroll") for (int i = 0; i < WM; ++i) {                                          \
            const int idx_ = tid + 256 * i;                                                       \
            if (idx_ < nw) wr[i] = wc_[(idx_ / BN) * p.cout_pad + (idx_ % BN)];                   \
        }                                                                                         \
    } while (0)

    // phase stamps (tools/bench_ops.py conv_stamps): wave 0 accumulates where its time goes
    const bool stamp = p.stamps != nullptr && wave == 0 && blockIdx.y == 0;
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
    CH_LOAD(0);
    for (int chunk = 0; chunk < p.nchunk; ++chunk) {
        // registers -> LDS (waits for this chunk's loads), converting fp32 sources to fp16
        if (PIV_DBG(p) & 4) {
        } else if (ld_f16) {
#pragma unroll
            for (int i = 0; i < PM; ++i)
                if (poff[i] != -2)
                    *reinterpret_cast<f32x4 *>(patch + ((tid >> 1) + 128 * i) * HPITCH + 8 * sub) = pr[i];
        } else {
#pragma unroll
            for (int i = 0; i < PM; ++i)
                if (poff[i] != -2) {
                    _Float16 *d = patch + ((tid >> 1) + 128 * i) * HPITCH + 4 * sub;
                    const f32x4 v0 = pr[2 * i], v1 = pr[2 * i + 1];
                    *reinterpret_cast<h4 *>(d) = h4{(_Float16)v0[0], (_Float16)v0[1], (_Float16)v0[2], (_Float16)v0[3]};
                    *reinterpret_cast<h4 *>(d + 8) = h4{(_Float16)v1[0], (_Float16)v1[1], (_Float16)v1[2], (_Float16)v1[3]};
                }
        }
        if (!(PIV_DBG(p) & 4)) {
#pragma unroll
            for (int i = 0; i < WM; ++i)
                if (tid + 256 * i < nw) reinterpret_cast<f32x4 *>(wts)[tid + 256 * i] = wr[i];
        }
        STAMP(d_commit);
        __syncthreads();
        STAMP(d_bar1);
        if (chunk + 1 < p.nchunk) {
            c0 += 16;
            if (c0 >= scl) {
                ++seg;
                c0 = 0;
                sp = reinterpret_cast<const char *>(p.seg[seg].ptr);
                scl = p.seg[seg].cload; sst = p.seg[seg].stride; sf16 = p.seg[seg].f16;
            }
            CH_LOAD(chunk + 1);
        }
        STAMP(d_issue);
        int tap = 0;
        for (int ky = 0; ky < ((PIV_DBG(p) & 1) ? 0 : p.KH); ++ky) {
            for (int kx = 0; kx < p.KW; ++kx, ++tap) {
                const int toff = (ky * PW + kx) * HPITCH;
                h8 a[MT], wq[NT];
#pragma unroll
                for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const h8 *>(patch + abase[m] + toff);
#pragma unroll
                for (int n = 0; n < NT; ++n) wq[n] = *reinterpret_cast<const h8 *>(wts + tap * 2 * BN * 8 + bbase + n * 256);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[n], a[m], acc[m][n], 0, 0, 0);   // A = channels, B = pixels
            }
        }
        STAMP(d_taps);
        __syncthreads();
        STAMP(d_bar2);
    }
#undef CH_LOAD
    unsigned long long t_loop_end = 0;
    if (stamp) t_loop_end = __builtin_readcyclecounter();

    // Epilogue: lane&31 = pixel, registers 4g..4g+3 = channels 8g + 4*hh + {0..3} (same D layout as the fp32 kernel).
    // The bias comes from LDS: with the whole register file in use the 32 per-store global bias loads of the fp32 kernel's
    // epilogue cannot be batched and serialise, one L2 round trip each (measured: 16 us per tile, a third of the kernel).
    {
        const int ox = x0 + row;
        const bool interior = x0 + 32 <= p.Wo && y0 + TH <= p.Ho && n0 + BN <= p.cout_store;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int oy = y0 + wave * MT + m;
            const bool pix_ok = interior || (oy < p.Ho && ox < p.Wo);
            const size_t pix = (size_t)(b * p.Ho + (oy < p.Ho ? oy : 0)) * p.Wo + (ox < p.Wo ? ox : 0);
            if (p.out_f16) {
                // fp16 output: a lane's 4 channels are only 8 bytes.  Lanes l and l+32 hold the two halves of the same
                // 8-channel group of the same pixel, so one v_permlane32_swap per register pairs them up: the lower lane
                // stores channels 8g..8g+7 of the even group, the upper lane those of the odd group -- 16-byte stores.
                _Float16 *orow = reinterpret_cast<_Float16 *>(p.out) + pix * p.out_stride;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        unsigned w[2][2];        // [group parity][dword] = this lane's 4 channels of groups 2gp and 2gp+1, as fp16
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int g = 2 * gp + e;
                            f32x4 v = {acc[m][n][4 * g + 0], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]};
                            v += *reinterpret_cast<const f32x4 *>(lbias + n * 32 + 8 * g + 4 * hh);
                            if (p.lrelu) {
                                v[0] = lrelu01(v[0]); v[1] = lrelu01(v[1]); v[2] = lrelu01(v[2]); v[3] = lrelu01(v[3]);
                            }
                            const h4 hv = h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                            __builtin_memcpy(w[e], &hv, 8);
                        }
                        // after the swaps: lower lanes hold (own even group, partner's even group), upper lanes (partner's odd, own odd)
                        const auto s0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
                        const int ch = n0 + n * 32 + 16 * gp + 8 * hh;      // first of this lane's 8 channels
                        if (!pix_ok || (!interior && ch >= p.cout_store)) continue;
                        const u32x4 q = {s0[0], s1[0], s0[1], s1[1]};
                        if (interior || ch + 8 <= p.cout_store) *reinterpret_cast<u32x4 *>(orow + ch) = q;
                        else *reinterpret_cast<u32x2 *>(orow + ch) = u32x2{q[0], q[1]};
                    }
                }
                continue;
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = n0 + n * 32 + 8 * g + 4 * hh;
                    if (!pix_ok || (!interior && ch >= p.cout_store)) continue;
                    f32x4 v = {acc[m][n][4 * g + 0], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]};
                    v += *reinterpret_cast<const f32x4 *>(lbias + (ch - n0));
                    if (p.lrelu) {
                        v[0] = lrelu01(v[0]); v[1] = lrelu01(v[1]); v[2] = lrelu01(v[2]); v[3] = lrelu01(v[3]);
                    }
                    *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(p.out) + pix * p.out_stride + ch) = v;
                }
            }
        }
    }
    if (stamp && lane == 0) {
        const unsigned long long t_end = __builtin_readcyclecounter();
        unsigned long long *o = p.stamps + (size_t)blockIdx.x * 8;
        o[0] = d_commit; o[1] = d_bar1; o[2] = d_issue; o[3] = d_taps; o[4] = d_bar2;
        o[5] = t_end - t_loop_end;       // epilogue
        o[6] = t_end - t_begin;          // whole workgroup
        o[7] = t_begin;
    }
#undef STAMP
}

template <int MT, int NT, int PM, int WM>
static int launch_h(const ConvParamsH &p, hipStream_t st)
{
    constexpr int TH = 4 * MT, BN = NT * 32;
    const int PH = (TH - 1) * p.S + p.KH, PW = 31 * p.S + p.KW;
    const size_t lds = ((size_t)PH * PW * HPITCH + (size_t)p.KH * p.KW * 2 * BN * 8) * sizeof(_Float16) + BN * sizeof(float);
    PIV_REQUIRE(lds <= 160 * 1024, "conv_f16: LDS tile of %zu bytes exceeds 160 KiB (k=%dx%d s=%d)", lds, p.KH, p.KW, p.S);
    PIV_REQUIRE(PH * PW * 2 <= 256 * PM && p.KH * p.KW * 2 * BN <= 256 * WM, "conv_f16: internal staging bound exceeded");
    static LdsAttr attr;
    if (int rc = ensure_dyn_lds(attr, reinterpret_cast<const void *>(conv_f16_kernel<MT, NT, PM, WM>), 160 * 1024)) return rc;
    const int tiles = cdiv(p.Wo, 32) * cdiv(p.Ho, TH) * p.B;
    dim3 grid(tiles, p.cout_pad / BN);
    hipLaunchKernelGGL((conv_f16_kernel<MT, NT, PM, WM>), grid, dim3(256), lds, st, p);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

int launch_conv_h(const ConvParamsH &p_in, hipStream_t st)
{
    ConvParamsH p = p_in;
    PIV_SET_DBG(p, PIV_KNOB(3));
    p.stamps = reinterpret_cast<unsigned long long *>(((unsigned long long)(unsigned)PIV_KNOB(6) << 32) | (unsigned)PIV_KNOB(5));   // tools only
    PIV_REQUIRE(p.nseg >= 1 && p.nseg <= 3 && p.nchunk >= 1, "conv_f16: bad segment description");
    PIV_REQUIRE(p.cout_pad % 32 == 0 && p.cout_store <= p.cout_pad && p.cout_store % 4 == 0, "conv_f16: bad output channel counts");
    for (int i = 0; i < p.nseg; ++i)
        PIV_REQUIRE(p.seg[i].ptr && p.seg[i].cload % (p.seg[i].f16 ? 8 : 4) == 0 && p.seg[i].stride % (p.seg[i].f16 ? 8 : 4) == 0,
                    "conv_f16: source %d must be 16-byte granular", i);
    PIV_REQUIRE(p.Ho > 0 && p.Wo > 0 && p.B > 0, "conv_f16: empty output");
    const bool s1_3x3 = p.S == 1 && p.KH <= 3 && p.KW <= 3;
    const int nt = (p.cout_pad % 128 == 0) ? 4 : (p.cout_pad % 64 == 0 ? 2 : 1);
    const long tiles16 = (long)cdiv(p.Wo, 32) * cdiv(p.Ho, 16) * p.B * (p.cout_pad / (32 * nt));
    const bool big = tiles16 >= 192;         // enough 16-row tiles to give (almost) every CU one
    if (s1_3x3) {
        if (nt == 4) return big ? launch_h<4, 4, 5, 9>(p, st) : launch_h<2, 4, 5, 9>(p, st);
        // 64 channels: the 8-row tile fits twice per CU (128->64 at 1024^2: 242 us vs 293 us for the 16-row tile)
        if (nt == 2) return (big && (PIV_KNOB(1) & 64)) ? launch_h<4, 2, 5, 9>(p, st) : launch_h<2, 2, 5, 9>(p, st);
        return big ? launch_h<4, 1, 5, 9>(p, st) : launch_h<2, 1, 5, 9>(p, st);
    }
    // everything else (stride 2, 7x7, separable k x 1 / 1 x k): 8-row tiles, the larger staging class
    if (nt == 4) return launch_h<2, 4, 9, 13>(p, st);
    if (nt == 2) return launch_h<2, 2, 9, 13>(p, st);
    return launch_h<2, 1, 9, 13>(p, st);
}

// OIHW fp32 weights -> fp16 [chunk][tap][k-half][cout_pad][8]; chunk = 16 staged input channels of one source.
void pack_conv_h(const float *w, int cout, int cin, int taps, const int *creal, const int *cload, const int *coff, int nseg,
                 std::vector<unsigned short> &pk, int *nchunk_out)
{
    const int cp = (cout + 31) / 32 * 32;
    int nchunk = 0;
    for (int s = 0; s < nseg; ++s) nchunk += (cload[s] + 15) / 16;
    pk.assign((size_t)nchunk * taps * 2 * cp * 8, 0);
    int chunk = 0, run = 0;
    for (int s = 0; s < nseg; ++s) {
        const int off = coff[s] >= 0 ? coff[s] : run;
        for (int c0 = 0; c0 < cload[s]; c0 += 16, ++chunk)
            for (int t = 0; t < taps; ++t)
                for (int kb = 0; kb < 2; ++kb)
                    for (int j = 0; j < 8; ++j) {
                        const int c = c0 + 8 * kb + j;
                        if (c >= creal[s]) continue;
                        for (int n = 0; n < cout; ++n) {
                            const _Float16 hv = (_Float16)w[((size_t)n * cin + off + c) * taps + t];
                            unsigned short bits;
                            memcpy(&bits, &hv, 2);
                            pk[((((size_t)chunk * taps + t) * 2 + kb) * cp + n) * 8 + j] = bits;
                        }
                    }
        run += creal[s];
    }
    *nchunk_out = nchunk;
}

}  // namespace pivlfn
