// Shared declarations for the gfx950 kernels of libpivlfn.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "../../include/pivlfn.h"

namespace pivlfn {

void set_error(const char *fmt, ...);

// Tuning knobs for in-process A/B measurements exist only in the tools build (libpivlfn_tools.so, -DPIVLFN_TOOLS, loaded
// by tools/ alone): 0 = warp_corr variant, 1 = conv variant bits, 2/3/7 = ablation masks, 5/6 = stamp buffer address.
// In the production library every knob is the compile-time constant 0: no process-global mutable state.
#ifdef PIVLFN_TOOLS
extern int g_knob[16];
#define PIV_KNOB(i) (pivlfn::g_knob[i])
#define PIV_DBG(p) ((p).dbg)            // a kernel's ablation mask (timing runs of the tools): a field of its parameter block
#define PIV_SET_DBG(p, v) ((p).dbg = (v))
#else
#define PIV_KNOB(i) 0
#define PIV_DBG(p) 0                    // production: the constant 0 -- no field in the parameter blocks, no test in the kernels
#define PIV_SET_DBG(p, v) ((void)0)
#endif

// Opt-in to > 64 KiB of dynamic LDS for kernel `fn`, once per (call site, device); thread-safe.  `slot` is a zero-initialised
// static at the call site.
struct LdsAttr { int bytes[64]; };
int ensure_dyn_lds(LdsAttr &slot, const void *fn, int bytes);

#define PIV_CHECK_HIP(expr)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            pivlfn::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return PIVLFN_ERR_HIP;                                                           \
        }                                                                                    \
    } while (0)

#define PIV_REQUIRE(cond, ...)                                                               \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            pivlfn::set_error(__VA_ARGS__);                                                  \
            return PIVLFN_ERR_ARG;                                                           \
        }                                                                                    \
    } while (0)

int device_cus();      // compute units of the current device (cached per device; 256 when the query fails)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Blocks b and b+8 share an XCD (round-robin dispatch over the 8 XCDs); give every XCD a contiguous
// run of logical tile ids so neighbouring tiles (shared halos) hit the same L2.  Bijective for any n.
__device__ __forceinline__ int xcd_remap(int bid, int n)
{
    const int q = n >> 3, r = n & 7;
    const int xcd = bid & 7, k = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

__device__ __forceinline__ float lrelu01(float v) { return v >= 0.f ? v : 0.1f * v; }

// Bilinear taps of a back-warp sample at (fx, fy): pixel indices (y*W+x) of the four neighbours, -1 when the
// neighbour lies outside the image (padding_mode='zeros', align_corners=True: src/models.py:32-35).
struct Taps { int o00, o01, o10, o11; float w00, w01, w10, w11; };
__device__ __forceinline__ Taps make_taps(float fx, float fy, int H, int W)
{
    Taps t;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float ax = fx - x0f, ay = fy - y0f;
    // clamp before the int conversion so wild flows cannot overflow; anything clamped is out of range anyway
    const int x0 = (int)fminf(fmaxf(x0f, -2.f), (float)W);
    const int y0 = (int)fminf(fmaxf(y0f, -2.f), (float)H);
    const bool xa = x0 >= 0 && x0 < W, xb = x0 + 1 >= 0 && x0 + 1 < W;
    const bool ya = y0 >= 0 && y0 < H, yb = y0 + 1 >= 0 && y0 + 1 < H;
    t.o00 = (xa && ya) ? y0 * W + x0 : -1;
    t.o01 = (xb && ya) ? y0 * W + x0 + 1 : -1;
    t.o10 = (xa && yb) ? (y0 + 1) * W + x0 : -1;
    t.o11 = (xb && yb) ? (y0 + 1) * W + x0 + 1 : -1;
    t.w00 = (1.f - ax) * (1.f - ay);
    t.w01 = ax * (1.f - ay);
    t.w10 = (1.f - ax) * ay;
    t.w11 = ax * ay;
    return t;
}

// ---- convolution (conv_mfma.hip) -------------------------------------------------------------------
struct ConvSeg {
    const float *ptr;   // NHWC base, channel offset already applied
    int cload;          // channels staged from this source, multiple of 4 (zero-weight lanes allowed)
    int stride;         // floats between consecutive pixels
};

struct ConvParams {
    ConvSeg seg[3];
    int nseg;
    const float *wpk;   // [nchunk][KH*KW][2][cout_pad][4]
    const float *bias;  // [cout_pad]
    float *out;         // NHWC, channel offset applied
    int out_stride;     // floats per output pixel
    int cout_store;     // channels written per pixel (>= real Cout; the extra ones are exact zeros)
    int cout_pad;       // multiple of 32, >= cout_store
    const float *res;   // optional residual added before the activation (same pixel grid as out)
    int res_stride;
    int B, H, W;        // input
    int Ho, Wo;         // output
    int KH, KW, S, padY, padX;
    int nchunk;         // K chunks of 8 input channels over all segments (the last one may be a 4-channel tail)
    int tail;           // 1 when the last chunk is a 4-channel tail (last source has cload % 8 == 4)
    int cin_real;       // real input channels of the layer (the weights of padding lanes are zero); 0 = not stated
    int lrelu;
    // split-K for layers with too few tiles to fill the chip (the coarse pyramid levels): gridDim.z workgroups share a tile,
    // each contracts a contiguous range of K chunks into scratch[z][pixel][cout_pad]; conv_splitk_reduce_kernel adds the
    // partial sums in z order (deterministic), then bias / residual / activation.  scratch == nullptr: never split.
    float *scratch;
    size_t scratch_floats;
    int ksplit;         // number of K shares (set by the launcher; 0/1 = no split)
    unsigned long long *stamps;   // tools only: per-workgroup phase times (s_memtime ticks), 8 per workgroup; nullptr in production
#ifdef PIVLFN_TOOLS
    int dbg;            // tools only (-DPIVLFN_STAMPS builds): ablation mask, 1 no MFMAs, 2 no global loads, 4 no epilogue stores
#endif
};

int launch_conv(const ConvParams &p, hipStream_t st);
// NetC.conv1 with the two 1 x 1 layers that read its output at level 1 -- NetC_ext (32 -> 64, both frames) and Regularization's
// moduleFeat (32 -> 128, first frame) -- computed from the activated accumulators in the same kernel (conv_mfma.hip).
struct Conv1Fuse {
    const float *w11;   // [6 blocks: 2 of NetC_ext, 4 of moduleFeat][4 groups][64 lanes][4]: W[32 blk + (lane & 31)][8 g + 4 (lane >> 5) + e]
    const float *b11;   // [64 + 128]
    float *out_ext;     // [images][H][W][64]
    float *out_feat;    // [first B_feat images][H][W][128]
    int B_feat;
};
int launch_conv1_fused(const ConvParams &p, const Conv1Fuse &f, hipStream_t st);     // -1: the layer / size is not covered
int launch_touch(const float *p, size_t floats, float *sink, hipStream_t st);     // read-only prefetch pass (ops.hip)
int launch_splitk_reduce(const ConvParams &p, int nz, hipStream_t st);     // second pass of a split-K layer (also used by conv_split.hip)

// ---- fp16-multiplicand variant (conv_f16.hip): optional reduced-precision mode, K chunks of 16 channels ----------------
struct ConvSegH {
    const void *ptr;    // NHWC base (fp32 or fp16 elements), channel offset already applied
    int cload;          // channels staged from this source: multiple of 4 (fp32) or 8 (fp16); zero-weight lanes allowed
    int stride;         // elements between consecutive pixels
    int f16;            // element type of this source: 0 = fp32 (converted while staging), 1 = fp16
};
struct ConvParamsH {
    ConvSegH seg[3];
    int nseg;
    const void *wpk;    // fp16 [nchunk][KH*KW][2][cout_pad][8]
    const float *bias;  // [cout_pad]
    void *out;          // NHWC fp32 or fp16
    int out_stride, cout_store, cout_pad, out_f16;
    int B, H, W, Ho, Wo;
    int KH, KW, S, padY, padX;
    int nchunk, lrelu;
#ifdef PIVLFN_TOOLS
    int dbg;            // ablation mask for tools/bench_ops.py (0 in production): 1 no MFMAs, 2 no global loads, 4 no LDS commit
#endif
    unsigned long long *stamps;   // tools only: per-workgroup phase times (s_memtime ticks), 8 per workgroup; nullptr in production
};
int launch_conv_h(const ConvParamsH &p, hipStream_t st);
// ---- fp32 by exact fp16 operand splitting (conv_split.hip): fp32 sources and output, K chunks of 16 channels ------------
struct ConvParamsX {
    ConvSeg seg[3];
    int nseg;
    const void *wpk;    // fp16 pieces [nchunk][KH*KW][3][2][cout_pad][8]
    const void *wtail;  // 3 x 3 layers whose staged channels end in a 4-lane tail: that chunk with taps folded into K,
                        // [step 3][piece 2][k-half 2][cout_pad][8 = 2 taps x 4 channels]; nullptr otherwise (conv_split_big_kernel)
    int tail;           // set by the launcher: the kernel uses wtail for the last chunk
    const float *bias;  // [cout_pad]
    float *out;         // NHWC fp32
    int out_stride, cout_store, cout_pad;
    float out_scale;    // 2^-k: undoes the power-of-two scale of the packed weights
    int terms;          // partial products per product: 6 (exact to 2^-32) or 3 (h.h, h.m, m.h: 2^-21)
    float *scratch;     // split-K scratch [z][pixel][cout_pad] (nullptr: never split); the policy is in launch_conv_x
    size_t scratch_floats;
    int ksplit;         // set by the launcher
    int B, H, W, Ho, Wo;
    int KH, KW, S, padY, padX;
    int nchunk, lrelu;
#ifdef PIVLFN_TOOLS
    int dbg;            // ablation mask for tools/bench_ops.py (0 in production): 1 no MFMAs, 2 no global loads, 4 no LDS commit
#endif
};
int launch_conv_x(const ConvParamsX &p, hipStream_t st);
bool conv_split_supports(int KH, int KW, int S, int cout_pad, int terms);
// ---- 3x3 stride-1 layers by Winograd F(2x2, 3x3) on the fp32 matrix cores (conv_wino.hip): fp32 everywhere, K chunks of 8 channels ----
struct ConvParamsW {
    ConvSeg seg[3];
    int nseg;
    const float *wpk;   // Winograd-domain weights, [nchunk][cout_pad/32][4][4][64][4] (pack_conv_w)
    const float *bias;  // [cout_pad]
    float *out;         // NHWC fp32, channel offset applied
    int out_stride, cout_store, cout_pad;
    int B, H, W;        // output grid = input grid (3x3, stride 1, pad 1)
    int nchunk, lrelu;
    unsigned long long *stamps;   // tools build (-DPIVLFN_STAMPS) only: per-workgroup phase times of wave 0 (s_memtime ticks), 8 per workgroup; nullptr in production
#ifdef PIVLFN_TOOLS
    int dbg;            // tools build only: ablation mask of conv_wino_ws.hip (timing runs, wrong results)
#endif
    const void *wpk_b;  // conv_wino_b3.hip: the same U split into three bf16 pieces, [nchunk][cout_pad/64][4][4][2][3][64][8] (pack_conv_wb); nchunk = K steps of 16 channels
    int terms;          // conv_wino_b3.hip: piece products per product, 6 (default), 8 or 9
};
int launch_conv_w(const ConvParamsW &p, hipStream_t st);
int launch_conv_wb(const ConvParamsW &p, hipStream_t st);      // the same layers with exactly split operands on the bf16 matrix cores (conv_wino_b3.hip)
bool conv_wino_b3_supports(int cout_pad);
int launch_conv_w_ws(const ConvParamsW &p, hipStream_t st);   // the same layers on persistent workgroups with producer / consumer waves (conv_wino_ws.hip); bit-identical
long conv_wino_ws_items(const ConvParamsW &p);                // work items such a launch would have (0: not covered)
int launch_conv_w4(const ConvParamsW &p, hipStream_t st);      // F(4x4, 3x3): wpk = [nchunk][cout_pad/32][6][6][64][4] (pack_conv_w4)
bool conv_wino_supports(int KH, int KW, int S, int padY, int padX);
// 32 -> 2 channel k x k flow head on the VALU (conv_head.hip); w = [k*k][8][2][4] on the device
// (7 x 1) convolution from 32 channels to up to 64 on v_mfma_f32_16x16x4_f32, operands from global memory (conv_head.hip)
int launch_conv_col7(const float *x, int x_stride, const float *wf, const float *bias, float *out, int out_stride, int cout_store,
                     int single48, int B, int H, int W, hipStream_t st);
// (1 x 7) convolution 49 -> 49 channels (52 stored lanes), the same way along x (conv_head.hip)
int launch_conv_row7(const float *x, int x_stride, const float *wf, const float *wf12, const float *bias, float *out, int out_stride,
                     int B, int H, int W, hipStream_t st);
int launch_conv_head(const float *x, const float *w, float b0, float b1, const float *res4, float *out4, int B, int H, int W,
                     int k, hipStream_t st);

// ---- warp + correlation (warp_corr.hip) ------------------------------------------------------------
int launch_warp_corr(const float *f1, const float *f2, const float *flow, float flow_scale, float *out,
                     int B, int C, int H, int W, int stride, int leaky, bool nhwc, hipStream_t st);
void warp_corr_time_next(hipEvent_t start, hipEvent_t stop);   // attach start/stop events to the next channels-last launch
int launch_backwarp_nchw(const float *in, const float *flow, float *out, int B, int C, int H, int W, hipStream_t st);
int launch_corr_bwd(const float *first, const float *second, const float *gout, float *gfirst, float *gsecond,
                    int B, int C, int H, int W, int s, hipStream_t st);

// ---- small ops (ops.hip); NHWC unless noted ----------------------------------------------------------
int launch_prep_images(const float *img1, const float *img2, float *out, int B, int H, int W,
                       const float mean[6], hipStream_t st);                       // NCHW x2 -> [2B,H,W,4]
int launch_resize_nhwc4(const float *in, float *out, int N, int H, int W, int Ho, int Wo, hipStream_t st);
int launch_resize_nchw(const float *in, float *out, int B, int C, int H, int W, int Ho, int Wo,
                       float m0, float m1, int use_mul, hipStream_t st);
int launch_dwconvT(const float *in, const float *w, float *out, int B, int H, int W, int C, int stride_in,
                   int stride_out, int cstore, hipStream_t st);                    // k4 s2 p1 depthwise
int launch_backwarp_nhwc(const float *in, const float *flow4, float scale, float *out, int B, int H, int W,
                         int C, hipStream_t st);
int launch_flow_mean(const float *flow4, float *partial, float *mean, int B, int HW, hipStream_t st);
int launch_reg_prep(const float *img1, const float *img2, const float *flow4, float *mean, const float *partial, float scale,
                    float *misc4, int B, int H, int W, hipStream_t st);
int launch_reg_tail(const float *dist, int dstride, const float *flow4, const float *wx, const float *wy,
                    float bx, float by, int k, float *out4, float *out_nchw, float out_scale,
                    int B, int H, int W, hipStream_t st);
int launch_flow4_to_nchw(const float *flow4, float *out, int B, int H, int W, hipStream_t st);
int flow_mean_partials(int HW);

}  // namespace pivlfn
