// extern "C" surface of libpivlfn.so (declared in include/pivlfn.h).
#include <atomic>
#include <mutex>
#include <vector>
#include "common.h"

struct pivlfn_net;
struct pivlfn_conv;

namespace pivlfn {

static thread_local char g_err[512] = "";
#ifdef PIVLFN_TOOLS
int g_knob[16] = {0};
#endif

// hipFuncSetAttribute is per device: the opt-in is remembered per (call site, device).  Fast path: one relaxed atomic read.
int ensure_dyn_lds(LdsAttr &slot, const void *fn, int bytes)
{
    static std::mutex mu;
    int dev = 0;
    PIV_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { set_error("device ordinal %d out of range", dev); return PIVLFN_ERR_ARG; }
    if (__atomic_load_n(&slot.bytes[dev], __ATOMIC_ACQUIRE) >= bytes) return PIVLFN_OK;
    std::lock_guard<std::mutex> lock(mu);
    if (slot.bytes[dev] >= bytes) return PIVLFN_OK;
    PIV_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    __atomic_store_n(&slot.bytes[dev], bytes, __ATOMIC_RELEASE);
    return PIVLFN_OK;
}

int device_cus()
{
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = __atomic_load_n(&cus[dev], __ATOMIC_ACQUIRE);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        __atomic_store_n(&cus[dev], n, __ATOMIC_RELEASE);
    }
    return n;
}

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int net_create(const pivlfn_tensor *tensors, int n, float starting_scale, int lowest, const float mean[6], pivlfn_net **out);
size_t net_workspace_bytes(const pivlfn_net *net, int B, int H, int W);
size_t net_levels_floats(const pivlfn_net *net, int B, int H, int W);
int net_forward(pivlfn_net *net, const float *img1, const float *img2, float *flow, float *levels, int B, int H, int W,
                void *ws, size_t ws_bytes, hipStream_t st);
int net_destroy(pivlfn_net *net);
int net_profile_enable(pivlfn_net *net, int level);
int net_profile_read(pivlfn_net *net, double *ms, double *ms_empty, long *launches, int reset);
int conv_create(const float *weight, const float *bias, int cout, int cin, int kh, int kw, pivlfn_conv **out);
int conv_destroy(pivlfn_conv *c);
int conv_forward(const pivlfn_conv *c, const float *x, int x_stride, float *y, int y_stride, const float *res, int res_stride,
                 int B, int H, int W, int stride, int pad_y, int pad_x, int leaky, hipStream_t st);
int conv_forward_h(const pivlfn_conv *c, const void *x, int x_stride, int x_f16, void *y, int y_stride, int y_f16,
                   int B, int H, int W, int stride, int pad_y, int pad_x, int leaky, hipStream_t st);
int net_set_precision(pivlfn_net *net, int precision);
int conv_forward_x(const pivlfn_conv *c, const float *x, int x_stride, float *y, int y_stride,
                   int B, int H, int W, int stride, int pad_y, int pad_x, int leaky, int terms, hipStream_t st);
int conv_head_forward(const pivlfn_conv *c, const float *x, const float *res4, float *out4, int B, int H, int W, hipStream_t st);
int conv_create_cat(const float *weight, const float *bias, int cout, int nsrc, const int *channels, int kh, int kw, pivlfn_conv **out);
int conv_forward_cat(const pivlfn_conv *c, int nsrc, const float *const *x, const int *x_stride, float *y, int y_stride,
                     int B, int H, int W, int leaky, hipStream_t st);
int conv_forward_wb(const pivlfn_conv *c, const float *x, int x_stride, float *y, int y_stride, int B, int H, int W, int leaky,
                    int terms, hipStream_t st);
int conv_forward_w(const pivlfn_conv *c, const float *x, int x_stride, float *y, int y_stride, int B, int H, int W, int leaky,
                   hipStream_t st, int tile);

}  // namespace pivlfn

using namespace pivlfn;

extern "C" {

const char *pivlfn_last_error(void) { return g_err; }
int pivlfn_abi_version(void) { return 3; }

#ifdef PIVLFN_TOOLS
int pivlfn_tune(int knob, int value)
{
    if (knob < 0 || knob >= 16) { set_error("tune: knob %d out of range", knob); return PIVLFN_ERR_ARG; }
    g_knob[knob] = value;
    return PIVLFN_OK;
}
#endif

int pivlfn_corr_fwd(const float *first, const float *second, float *out, int B, int C, int H, int W, int stride, void *stream)
{
    return launch_warp_corr(first, second, nullptr, 0.f, out, B, C, H, W, stride, 0, false, (hipStream_t)stream);
}

int pivlfn_corr_bwd(const float *first, const float *second, const float *grad_out, float *grad_first, float *grad_second,
                    int B, int C, int H, int W, int stride, void *stream)
{
    return launch_corr_bwd(first, second, grad_out, grad_first, grad_second, B, C, H, W, stride, (hipStream_t)stream);
}

int pivlfn_backwarp(const float *in, const float *flow, float *out, int B, int C, int H, int W, void *stream)
{
    return launch_backwarp_nchw(in, flow, out, B, C, H, W, (hipStream_t)stream);
}

int pivlfn_warp_corr_fwd(const float *first, const float *second, const float *flow, float flow_scale, float *out,
                         int B, int C, int H, int W, int stride, int leaky, void *stream)
{
    return launch_warp_corr(first, second, flow, flow_scale, out, B, C, H, W, stride, leaky, false, (hipStream_t)stream);
}

int pivlfn_warp_corr_nhwc(const float *first, const float *second, const float *flow, float flow_scale, float *out,
                          int B, int C, int H, int W, int stride, int leaky, void *stream)
{
    return launch_warp_corr(first, second, flow, flow_scale, out, B, C, H, W, stride, leaky, true, (hipStream_t)stream);
}

int pivlfn_resize_bilinear(const float *in, float *out, int B, int C, int H, int W, int Ho, int Wo, const float *mul,
                           void *stream)
{
    return launch_resize_nchw(in, out, B, C, H, W, Ho, Wo, mul ? mul[0] : 1.f, mul ? mul[1] : 1.f, mul ? 1 : 0,
                              (hipStream_t)stream);
}

int pivlfn_create(const pivlfn_tensor *tensors, int n_tensors, float starting_scale, int lowest_level,
                  const float rgb_mean[6], pivlfn_net **out)
{
    return net_create(tensors, n_tensors, starting_scale, lowest_level, rgb_mean, out);
}

int pivlfn_destroy(pivlfn_net *net) { return net_destroy(net); }

size_t pivlfn_workspace_bytes(const pivlfn_net *net, int B, int H, int W)
{
    if (!net || B <= 0 || H <= 0 || W <= 0) return 0;
    return net_workspace_bytes(net, B, H, W);
}

size_t pivlfn_levels_floats(const pivlfn_net *net, int B, int H, int W)
{
    if (!net || B <= 0 || H <= 0 || W <= 0) return 0;
    return net_levels_floats(net, B, H, W);
}

int pivlfn_forward(pivlfn_net *net, const float *img1, const float *img2, float *flow, float *levels, int B, int H, int W,
                   void *workspace, size_t workspace_bytes, void *stream)
{
    return net_forward(net, img1, img2, flow, levels, B, H, W, workspace, workspace_bytes, (hipStream_t)stream);
}

int pivlfn_conv_create(const float *weight, const float *bias, int cout, int cin, int kh, int kw, pivlfn_conv **out)
{
    return conv_create(weight, bias, cout, cin, kh, kw, out);
}

int pivlfn_conv_destroy(pivlfn_conv *conv) { return conv_destroy(conv); }

int pivlfn_conv2d_nhwc(const pivlfn_conv *conv, const float *x, int x_stride, float *y, int y_stride, const float *res,
                       int res_stride, int B, int H, int W, int stride, int pad_y, int pad_x, int leaky, void *stream)
{
    return conv_forward(conv, x, x_stride, y, y_stride, res, res_stride, B, H, W, stride, pad_y, pad_x, leaky, (hipStream_t)stream);
}

int pivlfn_conv2d_nhwc_f16(const pivlfn_conv *conv, const void *x, int x_stride, int x_is_f16, void *y, int y_stride,
                           int y_is_f16, int B, int H, int W, int stride, int pad_y, int pad_x, int leaky, void *stream)
{
    return conv_forward_h(conv, x, x_stride, x_is_f16, y, y_stride, y_is_f16, B, H, W, stride, pad_y, pad_x, leaky, (hipStream_t)stream);
}

int pivlfn_conv2d_nhwc_split(const pivlfn_conv *conv, const float *x, int x_stride, float *y, int y_stride,
                             int B, int H, int W, int stride, int pad_y, int pad_x, int leaky, int terms, void *stream)
{
    return conv_forward_x(conv, x, x_stride, y, y_stride, B, H, W, stride, pad_y, pad_x, leaky, terms, (hipStream_t)stream);
}

int pivlfn_conv2d_nhwc_wino(const pivlfn_conv *conv, const float *x, int x_stride, float *y, int y_stride,
                            int B, int H, int W, int leaky, void *stream)
{
    return conv_forward_w(conv, x, x_stride, y, y_stride, B, H, W, leaky, (hipStream_t)stream, 2);
}

int pivlfn_conv2d_nhwc_wino_b3(const pivlfn_conv *conv, const float *x, int x_stride, float *y, int y_stride,
                               int B, int H, int W, int leaky, int terms, void *stream)
{
    return conv_forward_wb(conv, x, x_stride, y, y_stride, B, H, W, leaky, terms, (hipStream_t)stream);
}

#ifdef PIVLFN_TOOLS
int pivlfn_conv2d_nhwc_wino4(const pivlfn_conv *conv, const float *x, int x_stride, float *y, int y_stride,
                             int B, int H, int W, int leaky, void *stream)
{
    return conv_forward_w(conv, x, x_stride, y, y_stride, B, H, W, leaky, (hipStream_t)stream, 4);
}
#endif

int pivlfn_conv_create_cat(const float *weight, const float *bias, int cout, int nsrc, const int *channels, int kh, int kw, pivlfn_conv **out)
{
    return conv_create_cat(weight, bias, cout, nsrc, channels, kh, kw, out);
}

int pivlfn_conv2d_nhwc_cat(const pivlfn_conv *conv, int nsrc, const float *const *x, const int *x_stride, float *y, int y_stride,
                           int B, int H, int W, int leaky, void *stream)
{
    return conv_forward_cat(conv, nsrc, x, x_stride, y, y_stride, B, H, W, leaky, (hipStream_t)stream);
}

int pivlfn_set_precision(pivlfn_net *net, int precision) { return net_set_precision(net, precision); }

int pivlfn_conv_head_nhwc(const pivlfn_conv *conv, const float *x, const float *res4, float *out4, int B, int H, int W, void *stream)
{
    return conv_head_forward(conv, x, res4, out4, B, H, W, (hipStream_t)stream);
}

int pivlfn_warp_corr_nhwc_timed(const float *first, const float *second, const float *flow, float flow_scale, float *out,
                                int B, int C, int H, int W, int stride, int leaky, int launches, double *us_dispatch, void *stream)
{
    PIV_REQUIRE(launches >= 1 && launches <= 256 && us_dispatch, "warp_corr_nhwc_timed: launches=%d must be 1..256, us_dispatch non-null", launches);
    hipStream_t st = (hipStream_t)stream;
    std::vector<hipEvent_t> ev(2 * (size_t)launches, nullptr);
    int rc = PIVLFN_OK;
    for (auto &e : ev)
        if (hipEventCreate(&e) != hipSuccess) { set_error("hipEventCreate failed"); rc = PIVLFN_ERR_HIP; break; }
    for (int i = 0; rc == PIVLFN_OK && i < launches; ++i) {      // back to back: every dispatch carries its own start / stop events
        warp_corr_time_next(ev[2 * i], ev[2 * i + 1]);
        rc = launch_warp_corr(first, second, flow, flow_scale, out, B, C, H, W, stride, leaky, true, st);
    }
    warp_corr_time_next(nullptr, nullptr);
    double total = 0.0;
    if (rc == PIVLFN_OK && hipStreamSynchronize(st) != hipSuccess) { set_error("hipStreamSynchronize failed"); rc = PIVLFN_ERR_HIP; }
    for (int i = 0; rc == PIVLFN_OK && i < launches; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) != hipSuccess) { set_error("hipEventElapsedTime failed"); rc = PIVLFN_ERR_HIP; break; }
        total += ms;
    }
    for (auto &e : ev) if (e) (void)hipEventDestroy(e);
    if (rc == PIVLFN_OK) *us_dispatch = total * 1e3 / launches;
    return rc;
}

int pivlfn_profile_enable(pivlfn_net *net, int level) { return net_profile_enable(net, level); }

int pivlfn_profile_read(pivlfn_net *net, double *ms_total, double *ms_empty_pairs, long *launches, int reset)
{
    return net_profile_read(net, ms_total, ms_empty_pairs, launches, reset);
}

}  // extern "C"
