/*
 * pivlfn.h -- C ABI of libpivlfn.so: the MI355X (gfx950) PIV-LiteFlowNet inference hot path.
 *
 * This is the drop-in boundary for the reference's src/models.py + src/correlation.py.  Every entry
 * point takes plain device pointers, sizes and a hipStream_t (passed as void*); nothing here
 * allocates per call (outputs and the workspace are the caller's), every function returns 0 on
 * success or a non-zero code, and pivlfn_last_error() gives the thread-local message.
 * All tensors are fp32.  "NCHW" tensors are contiguous, exactly what the reference's Python passes.
 * Citations are into /root/reference/.
 */
#ifndef PIVLFN_H
#define PIVLFN_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIVLFN_OK            0
#define PIVLFN_ERR_ARG       1   /* bad shape / null pointer / unsupported configuration */
#define PIVLFN_ERR_HIP       2   /* a HIP runtime call failed (message has the hipError string) */
#define PIVLFN_ERR_WORKSPACE 3   /* workspace too small */
#define PIVLFN_ERR_WEIGHTS   4   /* state dict does not match the network layout */

typedef struct pivlfn_net pivlfn_net;   /* opaque: packed weights of one network on one device */

/* One entry of a state dict (host memory, fp32, contiguous).  Same names and shapes as
 * LiteFlowNet.state_dict() in src/models.py:305-317 (key pattern in SURVEY.md section 8 a10). */
typedef struct {
    const char  *name;
    const float *data;
    int          ndim;
    int          shape[4];
} pivlfn_tensor;

const char *pivlfn_last_error(void);
/* ABI version.  3 (round 6): + pivlfn_conv2d_nhwc_wino_b3, PIVLFN_PRECISION_F32_WINO_MFMA32; - pivlfn_conv2d_nhwc_wino4 (now exported by the tools build only).  2 (round 4): + pivlfn_warp_corr_nhwc_timed, pivlfn_conv2d_nhwc_wino4, pivlfn_conv_create_cat, pivlfn_conv2d_nhwc_cat; since 1 also pivlfn_conv2d_nhwc_wino and PIVLFN_PRECISION_F32_DIRECT
 * (added in round 3 without a bump).  No entry point of version 1 changed its signature or meaning. */
int         pivlfn_abi_version(void);

/* The library keeps no process-global mutable state: entry points may be called concurrently from several threads, on
 * several devices and streams (one pivlfn_net / pivlfn_conv handle per device; a handle is used by one thread at a time).
 * Kernel-variant knobs for in-process A/B timing exist only in the separate tools build (libpivlfn_tools.so, compiled with
 * -DPIVLFN_TOOLS and loaded by tools/ alone); libpivlfn.so does not export pivlfn_tune. */
#ifdef PIVLFN_TOOLS
int         pivlfn_tune(int knob, int value);
#endif

/* ---- custom op: replaces _FunctionCorrelation.forward, src/correlation.py:287-344 (+ kernels :9-104)
 * first, second: NCHW [B,C,H,W]; out: NCHW [B,49,ceil(H/stride),ceil(W/stride)];
 * out[b,7(dy+3)+(dx+3),y,x] = (1/C) sum_c first[b,c,s*y,s*x] * second[b,c,s*(y+dy),s*(x+dx)], zeros outside. */
int pivlfn_corr_fwd(const float *first, const float *second, float *out,
                    int B, int C, int H, int W, int stride, void *stream);

/* ---- backward of the custom op: replaces _FunctionCorrelation.backward, src/correlation.py:348-405 (+ kernels :106-234).
 * grad_out: NCHW [B,49,ceil(H/stride),ceil(W/stride)]; grad_first / grad_second: NCHW [B,C,H,W], every element written
 * (exact zeros off the stride grid); either may be NULL (needs_input_grad false, :353-356). */
int pivlfn_corr_bwd(const float *first, const float *second, const float *grad_out, float *grad_first, float *grad_second,
                    int B, int C, int H, int W, int stride, void *stream);

/* ---- replaces backwarp(), src/models.py:20-35.  in: NCHW [B,C,H,W]; flow: NCHW [B,2,H,W] (pixels);
 * out[b,c,y,x] = bilinear(in[b,c], x + flow[b,0,y,x], y + flow[b,1,y,x]), zeros outside. */
int pivlfn_backwarp(const float *in, const float *flow, float *out,
                    int B, int C, int H, int W, void *stream);

/* ---- fused Matching front end: backwarp(second, flow*flow_scale) then correlation, then optional
 * LeakyReLU(0.1): src/models.py:171-184.  NCHW in / NCHW out, flow may be NULL (level 6). */
int pivlfn_warp_corr_fwd(const float *first, const float *second, const float *flow, float flow_scale,
                         float *out, int B, int C, int H, int W, int stride, int leaky, void *stream);

/* ---- the same kernel on the network's internal channels-last layout (what pivlfn_forward launches;
 * exported for benchmarks and roofline measurement).  first/second: [B,H,W,C]; flow: [B,H,W,4] (u,v,0,0)
 * or NULL; out: [B,Ho,Wo,56] (49 displacements + 7 zero lanes). C must be a multiple of 32. */
int pivlfn_warp_corr_nhwc(const float *first, const float *second, const float *flow, float flow_scale,
                          float *out, int B, int C, int H, int W, int stride, int leaky, void *stream);

/* ---- measurement hook: `launches` (1..256) back-to-back launches of pivlfn_warp_corr_nhwc on `stream`, each dispatch carrying
 * its own start / stop events (hipExtLaunchKernelGGL: the dispatch's begin / end timestamps, the figure rocprofv3 reports per
 * kernel -- no inter-kernel gap, no marker packets).  Synchronises the stream; *us_dispatch = mean microseconds per dispatch. */
int pivlfn_warp_corr_nhwc_timed(const float *first, const float *second, const float *flow, float flow_scale,
                                float *out, int B, int C, int H, int W, int stride, int leaky, int launches,
                                double *us_dispatch, void *stream);

/* ---- bilinear resize, align_corners=False, NCHW -> NCHW, with a per-channel multiplier
 * (mul[c % 2] when mul != NULL, host pointer to 2 floats): the two interpolate calls and the flow
 * rescale of estimate(), inference.py:46-49 and :57-61. */
int pivlfn_resize_bilinear(const float *in, float *out, int B, int C, int H, int W, int Ho, int Wo,
                           const float *mul, void *stream);

/* ---- network: replaces LiteFlowNet.__init__ + load_state_dict (src/models.py:39-317, 736-738, 762-764).
 * Uploads and repacks the weights once (this is the only call that allocates device memory).
 * starting_scale / lowest_level / rgb_mean as in the factories src/models.py:729-730, 754-755. */
int pivlfn_create(const pivlfn_tensor *tensors, int n_tensors, float starting_scale, int lowest_level,
                  const float rgb_mean[6], pivlfn_net **out);
int pivlfn_destroy(pivlfn_net *net);

/* Bytes of scratch pivlfn_forward needs for a [B,3,H,W] pair batch (H, W multiples of 32). */
size_t pivlfn_workspace_bytes(const pivlfn_net *net, int B, int H, int W);

/* ---- replaces LiteFlowNet.forward in eval mode, src/models.py:319-370.
 * img1, img2: NCHW [B,3,H,W] in [0,1] (NOT modified: the reference's in-place mean subtraction
 * :321-323 happens on an internal copy).  flow: NCHW [B,2,H/2^(lowest_level-1),W/2^(lowest_level-1)],
 * already multiplied by SCALEFACTOR[1] (:370).
 * levels (optional, may be NULL): receives the per-level [M,S,R] flows of the training-mode return
 * (:363-367), coarsest level first, each NCHW [B,2,h,w], packed back to back. */
int pivlfn_forward(pivlfn_net *net, const float *img1, const float *img2, float *flow, float *levels,
                   int B, int H, int W, void *workspace, size_t workspace_bytes, void *stream);

/* Precision of the conv stacks inside pivlfn_forward.
 * PIVLFN_PRECISION_F32 (the library's default, the mode every fp32 parity statement and the headline benchmark refer to):
 * fp32 operands, fp32 products and fp32 accumulation on the fp32 matrix-core instruction v_mfma_f32_32x32x2_f32 (exact fma
 * chains).  The 3 x 3 / stride 1 layers with an output grid of at least 64 x 64 per image are computed by Winograd's minimal
 * filtering F(2x2, 3x3) (csrc/conv_wino.hip; the algorithm cuDNN / MIOpen choose for fp32 3 x 3 layers: 2.25 x fewer multiplies,
 * all of them fp32 x fp32 on 24-bit operands; tests/test_gpu_wino.py measures the error against float64 next to the direct
 * kernel's); every other layer by direct convolution.  PIVLFN_PRECISION_F32_DIRECT: direct convolution for every layer.
 * PIVLFN_PRECISION_F16 (BASELINE config #5): operands rounded to fp16 while they are staged, fp32 accumulation, activations
 * still fp32 in HBM; flows agree with the fp32 mode to an end-point error stated in tests/test_gpu_f16.py.
 * Everything that is not a convolution (correlation, warps, flow heads, regularisation tail) is fp32 in all modes. */
#define PIVLFN_PRECISION_F32 0
#define PIVLFN_PRECISION_F16 1
#define PIVLFN_PRECISION_F32_DIRECT 4
/* PIVLFN_PRECISION_F32_WINO_MFMA32: the default of rounds 3-5 -- as PIVLFN_PRECISION_F32, but every Winograd layer on the fp32 matrix
 * instruction (csrc/conv_wino.hip).  Since round 6 PIVLFN_PRECISION_F32 runs the Winograd layers with whole 64-channel output groups
 * and >= 64 staged input channels on csrc/conv_wino_b3.hip: the same algorithm and the same fp32 U = G g G^T, each fp32 operand split
 * EXACTLY into three bf16 pieces (8 + 8 + 8 significand bits, fp32's exponent range: no narrower input domain) and each product
 * formed from six exact bf16 x bf16 products accumulated in fp32 on v_mfma_f32_32x32x16_bf16 -- what is dropped is <= 2^-23 of a
 * product (2^-26 typically), below the rounding of an fp32 fma; measured against float64 the layer error is at or below both
 * fp32-instruction kernels' (tests/test_gpu_wino_b3.py). */
#define PIVLFN_PRECISION_F32_WINO_MFMA32 5
/* PIVLFN_PRECISION_F32_SPLIT: fp32 results from the fp16 matrix cores.  Every fp32 operand is split exactly into three fp16
 * pieces (11 + 11 + 2 significand bits at scales 1, 2^-11, 2^-22) and each product is formed from six exact fp16 x fp16
 * products accumulated in fp32; what is dropped is below 2^-32 of a product, 256 x under the rounding of an fp32 fma
 * (csrc/conv_split.hip; tests/test_gpu_split.py measures the error against float64 next to the fp32 instruction's).
 * Applies to the residual-free stride-1 convolutions with an output grid of at least 256 x 256 per image; everything else runs
 * as in F32.  Inputs of those layers must stay below 65504 in magnitude (fp16 range of the leading piece; beyond it the
 * result is NaN, not a silently saturated value); inputs below 2^-14 in magnitude are represented to an absolute 2^-37
 * (2^-26 in SPLIT3) instead of exactly. */
#define PIVLFN_PRECISION_F32_SPLIT 2
/* PIVLFN_PRECISION_F32_SPLIT3: the same with two pieces per operand and the three leading partial products (h.h, h.m, m.h):
 * a product carries a relative error of at most 2^-21 (typically 2^-23.5, about one fp32 ulp on each operand); measured against
 * float64 the layer outputs' mean error stays within 2 x the fp32 instruction's (what tests/test_gpu_split.py asserts; on the layers
 * measured there it was lower: the fp16 instruction rounds once per 16 products), at half the matrix work of SPLIT.  Applies to the residual-free convolutions with an output grid of at least 64 x 64 per
 * image: stride 1 (4-row tiles and split-K on the small grids) and 3 x 3 stride 2.  Opt-in (not the default): the multiplicands
 * are 22-23 bits wide, one fewer than fp32's. */
#define PIVLFN_PRECISION_F32_SPLIT3 3
int pivlfn_set_precision(pivlfn_net *net, int precision);

/* Number of floats `levels` must hold for pivlfn_forward. */
size_t pivlfn_levels_floats(const pivlfn_net *net, int B, int H, int W);

/* ---- one convolution layer on the network's channels-last layout (what pivlfn_forward launches for every
 * torch.nn.Conv2d of src/models.py:70-106, 124, 154-163, 197-207, 229-272); exported so the kernel can be checked and
 * timed on its own.  weight: host, OIHW [cout,cin,kh,kw]; bias: host [cout].
 * x: [B,H,W,x_stride] (first cin lanes used, x_stride % 4 == 0, lanes cin..roundup(cin,4) must be finite);
 * y: [B,Ho,Wo,y_stride], lanes cout..min(roundup(cout,4), y_stride) are written as exact zeros;
 * res (optional): same grid as y, added before the activation; leaky: LeakyReLU(0.1) on the result.
 * Dispatch = the PIVLFN_PRECISION_F32 network's for the shape, except Winograd (own entry point below): the direct kernel
 * everywhere, but a 7 x 1 layer (pad 3, 0; no residual, no activation) on an image of >= 256 x 256 pixels runs on the streaming
 * matrix-core kernel pivlfn_forward uses for conv_dist_R.0 there -- not bit-comparable with the F32_DIRECT network's layer. */
typedef struct pivlfn_conv pivlfn_conv;
int pivlfn_conv_create(const float *weight, const float *bias, int cout, int cin, int kh, int kw, pivlfn_conv **out);
int pivlfn_conv_destroy(pivlfn_conv *conv);
int pivlfn_conv2d_nhwc(const pivlfn_conv *conv, const float *x, int x_stride, float *y, int y_stride,
                       const float *res, int res_stride, int B, int H, int W, int stride, int pad_y, int pad_x,
                       int leaky, void *stream);
/* The same layer in the optional reduced-precision mode (BASELINE config #5: fp16 multiplicands, fp32 accumulation, on
 * v_mfma_f32_32x32x16_f16): x is fp32 or fp16 elements (x_is_f16; stride granularity 4 / 8 elements), y is stored as fp32
 * or fp16 (y_is_f16).  No residual input. */
int pivlfn_conv2d_nhwc_f16(const pivlfn_conv *conv, const void *x, int x_stride, int x_is_f16, void *y, int y_stride,
                           int y_is_f16, int B, int H, int W, int stride, int pad_y, int pad_x, int leaky, void *stream);

/* The same layer (stride 1, no residual) on the split-operand kernel of PIVLFN_PRECISION_F32_SPLIT (terms = 6) or
 * PIVLFN_PRECISION_F32_SPLIT3 (terms = 3): fp32 x, fp32 y. */
int pivlfn_conv2d_nhwc_split(const pivlfn_conv *conv, const float *x, int x_stride, float *y, int y_stride,
                             int B, int H, int W, int stride, int pad_y, int pad_x, int leaky, int terms, void *stream);

/* The same layer (3 x 3, stride 1, pad 1, no residual) on the Winograd F(2x2, 3x3) kernel PIVLFN_PRECISION_F32 uses: fp32 x,
 * fp32 y, output grid = input grid. */
int pivlfn_conv2d_nhwc_wino(const pivlfn_conv *conv, const float *x, int x_stride, float *y, int y_stride,
                            int B, int H, int W, int leaky, void *stream);
/* The same layer (3 x 3, stride 1, pad 1, no residual; cout rounded up to 32 a multiple of 64) by Winograd F(2x2, 3x3) with every
 * fp32 operand split exactly into three bf16 pieces (8 + 8 + 8 significand bits, fp32's exponent range) and the products formed
 * on v_mfma_f32_32x32x16_bf16 (csrc/conv_wino_b3.hip): terms = 6 (dropped piece products <= 2^-23 of a product, typically 2^-26), 8
 * (<= 2^-32) or 9 (the exact product of the two fp32 operands).  fp32 x, fp32 y. */
int pivlfn_conv2d_nhwc_wino_b3(const pivlfn_conv *conv, const float *x, int x_stride, float *y, int y_stride,
                               int B, int H, int W, int leaky, int terms, void *stream);
#ifdef PIVLFN_TOOLS
/* Tools build only (tools/libpivlfn_tools.so; the kernel lives in tools/kernels/conv_wino4.hip since round 6): the same layer on the
 * Winograd F(4x4, 3x3) kernel (6 x 6 transforms; relative error ~1e-5 against ~1.4e-6 of F(2x2)).  Measured 0.98x of F(2x2) on
 * 128->128 at 1024 x 1024 and slower below: no user path launches it. */
int pivlfn_conv2d_nhwc_wino4(const pivlfn_conv *conv, const float *x, int x_stride, float *y, int y_stride,
                             int B, int H, int W, int leaky, void *stream);
#endif

/* One Conv2d (odd k, stride 1, "same" padding, + bias, optional LeakyReLU(0.1)) over the channel concatenation of 1-3 sources --
 * torch.cat + Conv2d of the front layers of Matching / Subpixel / Regularization (src/models.py:171-187, 209-217, 280) -- through
 * the dispatch of PIVLFN_PRECISION_F32: the multi-source staging of the direct and the Winograd kernel, for per-layer checks.
 * weight is OIHW over the concatenated channels; channels[i] = real channels of source i; x[i] is [B,H,W,x_stride[i]] with
 * x_stride[i] a multiple of 4 and >= channels[i] rounded up to 4 (padding lanes zero); only the last source may have a channel
 * count that is 4 (mod 8) after rounding. */
int pivlfn_conv_create_cat(const float *weight, const float *bias, int cout, int nsrc, const int *channels, int kh, int kw,
                           pivlfn_conv **out);
int pivlfn_conv2d_nhwc_cat(const pivlfn_conv *conv, int nsrc, const float *const *x, const int *x_stride, float *y, int y_stride,
                           int B, int H, int W, int leaky, void *stream);

/* The 32 -> 2 channel k x k flow head (conv_M.6 / conv_S.6) on its dedicated kernel: x [B,H,W,32], res4/out4 [B,H,W,4]. */
int pivlfn_conv_head_nhwc(const pivlfn_conv *conv, const float *x, const float *res4, float *out4, int B, int H, int W,
                          void *stream);

/* ---- measurement hooks.  With profiling on, pivlfn_forward times the level-`level` warp+correlation launch
 * with HIP events on `stream` in two ways: start/stop events attached to the dispatch itself
 * (hipExtLaunchKernelGGL: the dispatch's own begin/end timestamps) and a plain hipEventRecord pair around it
 * (which also contains the marker packets' own cost).  pivlfn_profile_read() synchronises those events and
 * returns both accumulated times in milliseconds and the launch count since the last reset. */
int pivlfn_profile_enable(pivlfn_net *net, int level);   /* level 1..6, 0 = off */
int pivlfn_profile_read(pivlfn_net *net, double *ms_dispatch, double *ms_event_pair, long *launches, int reset);

#ifdef __cplusplus
}
#endif
#endif /* PIVLFN_H */
