// HBM-bound glue kernels of the LiteFlowNet level pipeline (gfx950), channels-last (NHWC) fp32.
// Every kernel moves whole 16-byte lanes per thread and fuses what the reference runs as separate torch ops.
// Citations are into /root/reference/src/models.py.
#include "common.h"

namespace pivlfn {

using f32x4 = __attribute__((ext_vector_type(4))) float;

static inline int grid_for(size_t n, int cap = 16384)
{
    size_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > (size_t)cap ? cap : g));
}

// ---- per-axis source index / weights of torch's bilinear, align_corners=False -----------------------------
struct Lin { int i0, i1; float w0, w1; };
__device__ __forceinline__ Lin lin_src(int d, float scale, int n)
{
    float src = scale * ((float)d + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    Lin l;
    l.i0 = (int)src;
    if (l.i0 > n - 1) l.i0 = n - 1;
    l.i1 = l.i0 + (l.i0 < n - 1 ? 1 : 0);
    l.w1 = src - (float)l.i0;
    l.w0 = 1.f - l.w1;
    return l;
}

// ---- input: mean subtraction (:321-323) + NCHW -> [2B,H,W,4] (img1 batch then img2 batch) -------------------
__global__ __launch_bounds__(256) void prep_images_kernel(const float *__restrict__ img1, const float *__restrict__ img2,
                                                          float *__restrict__ out, int B, int HW,
                                                          float m0, float m1, float m2, float m3, float m4, float m5)
{
    const size_t total = (size_t)2 * B * HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int n = (int)(i / HW);
        const int pix = (int)(i - (size_t)n * HW);
        const bool second = n >= B;
        const float *src = (second ? img2 : img1) + (size_t)(second ? n - B : n) * 3 * HW + pix;
        f32x4 v;
        v[0] = src[0] - (second ? m3 : m0);
        v[1] = src[HW] - (second ? m4 : m1);
        v[2] = src[2 * (size_t)HW] - (second ? m5 : m2);
        v[3] = 0.f;
        reinterpret_cast<f32x4 *>(out)[i] = v;
    }
}

int launch_prep_images(const float *img1, const float *img2, float *out, int B, int H, int W, const float mean[6],
                       hipStream_t st)
{
    const size_t total = (size_t)2 * B * H * W;
    hipLaunchKernelGGL(prep_images_kernel, dim3(grid_for(total)), dim3(256), 0, st, img1, img2, out, B, H * W,
                       mean[0], mean[1], mean[2], mean[3], mean[4], mean[5]);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

// ---- image pyramid (:336-343): bilinear, align_corners=False, [N,H,W,4] -> [N,Ho,Wo,4] ----------------------
__global__ __launch_bounds__(256) void resize_nhwc4_kernel(const f32x4 *__restrict__ in, f32x4 *__restrict__ out, int N,
                                                           int H, int W, int Ho, int Wo, float sy, float sx)
{
    const size_t total = (size_t)N * Ho * Wo;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ox = (int)(i % Wo);
        const size_t r = i / Wo;
        const int oy = (int)(r % Ho);
        const int n = (int)(r / Ho);
        const Lin ly = lin_src(oy, sy, H), lx = lin_src(ox, sx, W);
        const f32x4 *base = in + (size_t)n * H * W;
        const f32x4 a = base[(size_t)ly.i0 * W + lx.i0], b = base[(size_t)ly.i0 * W + lx.i1];
        const f32x4 c = base[(size_t)ly.i1 * W + lx.i0], d = base[(size_t)ly.i1 * W + lx.i1];
        out[i] = ly.w0 * (lx.w0 * a + lx.w1 * b) + ly.w1 * (lx.w0 * c + lx.w1 * d);
    }
}

int launch_resize_nhwc4(const float *in, float *out, int N, int H, int W, int Ho, int Wo, hipStream_t st)
{
    const size_t total = (size_t)N * Ho * Wo;
    hipLaunchKernelGGL(resize_nhwc4_kernel, dim3(grid_for(total)), dim3(256), 0, st,
                       reinterpret_cast<const f32x4 *>(in), reinterpret_cast<f32x4 *>(out), N, H, W, Ho, Wo,
                       (float)H / (float)Ho, (float)W / (float)Wo);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

// ---- estimate()'s interpolate calls (inference.py:46-49, 57-61): NCHW, optional per-channel multiplier -------
__global__ __launch_bounds__(256) void resize_nchw_kernel(const float *__restrict__ in, float *__restrict__ out, int BC,
                                                          int H, int W, int Ho, int Wo, float sy, float sx,
                                                          float m0, float m1, int use_mul)
{
    const size_t total = (size_t)BC * Ho * Wo;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ox = (int)(i % Wo);
        const size_t r = i / Wo;
        const int oy = (int)(r % Ho);
        const int n = (int)(r / Ho);
        const Lin ly = lin_src(oy, sy, H), lx = lin_src(ox, sx, W);
        const float *base = in + (size_t)n * H * W;
        const float a = base[(size_t)ly.i0 * W + lx.i0], b = base[(size_t)ly.i0 * W + lx.i1];
        const float c = base[(size_t)ly.i1 * W + lx.i0], d = base[(size_t)ly.i1 * W + lx.i1];
        float v = ly.w0 * (lx.w0 * a + lx.w1 * b) + ly.w1 * (lx.w0 * c + lx.w1 * d);
        if (use_mul) v *= (n & 1) ? m1 : m0;
        out[i] = v;
    }
}

int launch_resize_nchw(const float *in, float *out, int B, int C, int H, int W, int Ho, int Wo, float m0, float m1,
                       int use_mul, hipStream_t st)
{
    PIV_REQUIRE(in && out && B > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "resize: bad arguments");
    PIV_REQUIRE(!use_mul || C % 2 == 0, "resize: per-channel multiplier needs an even channel count");
    const size_t total = (size_t)B * C * Ho * Wo;
    hipLaunchKernelGGL(resize_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, st, in, out, B * C, H, W, Ho, Wo,
                       (float)H / (float)Ho, (float)W / (float)Wo, m0, m1, use_mul);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

// ---- upConv_M / upCorr_M (:144-145, 151-152): depthwise ConvTranspose2d k4 s2 p1, no bias ----------------------
// out[oy,ox,c] = sum over the (at most 2x2) input pixels with oy = 2*iy - 1 + ky, ox = 2*ix - 1 + kx.
// w: [16 taps][C4] (zero lanes for padding channels, so padding lanes stay exact zeros); 32-bit index math.
// One workgroup row segment: blockIdx = (segment of 256 (pixel, quad) items along an output row, output row, image), Q a compile-time
// constant -- no run-time division anywhere (the flat-index form spent more on i % Q, / Wo, % Ho than on its four loads).
template <int Q>
__global__ __launch_bounds__(256) void dwconvT_kernel(const float *__restrict__ in, const float *__restrict__ w,
                                                      float *__restrict__ out, int H, int W, int sin, int sout)
{
    const int Ho = 2 * H, Wo = 2 * W;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int ox = j / Q, q = j - ox * Q;
    if (ox >= Wo) return;
    const int oy = blockIdx.y, b = blockIdx.z;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int ky0 = (oy + 1) & 1, kx0 = (ox + 1) & 1;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int ky = ky0 + 2 * a;
        const int iy = (oy + 1 - ky) >> 1;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int kx = kx0 + 2 * c;
            const int ix = (ox + 1 - kx) >> 1;
            if (ix < 0 || ix >= W) continue;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(in + ((size_t)(b * H + iy) * W + ix) * sin + 4 * q);
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + ((ky * 4 + kx) * Q + q) * 4);
            acc[0] = fmaf(v[0], wv[0], acc[0]);
            acc[1] = fmaf(v[1], wv[1], acc[1]);
            acc[2] = fmaf(v[2], wv[2], acc[2]);
            acc[3] = fmaf(v[3], wv[3], acc[3]);
        }
    }
    *reinterpret_cast<f32x4 *>(out + ((size_t)(b * Ho + oy) * Wo + ox) * sout + 4 * q) = acc;
}

// Read-only pass over a buffer: pulls it back into the Infinity Cache (and discards the values).  net.hip runs it on the side
// stream over the level-3 features once the side stream's 1.5 GB of 1x1-conv output has gone by (see there).
__global__ __launch_bounds__(256) void touch_kernel(const f32x4 *__restrict__ p, size_t n, float *__restrict__ sink)
{
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a += p[i];
    if (a[0] + a[1] + a[2] + a[3] == 1.2345678e38f) sink[0] = a[0];       // never true in practice: keeps the loads alive
}

// The same layer with a 2 x 2 output block per thread (Q > 1: the 49-channel correlation at levels 2 and 1, 235 MB of output at 1024²):
// the outputs (2m+1 .. 2m+2) x (2n+1 .. 2n+2) all read the input pixels (m .. m+1) x (n .. n+1), so a thread loads four input quads
// instead of sixteen and its sixteen weight quads once.  Every output sums its taps in dwconvT_kernel's order (input row m+1 before m,
// column n+1 before n, out-of-range taps skipped): the bits are the same.  Blocks m = -1 .. H-1, n = -1 .. W-1; blockIdx =
// (segment of 256 (block column, quad) items, block row + 1, image).
template <int Q>
__global__ __launch_bounds__(256) void dwconvT_b2_kernel(const float *__restrict__ in, const float *__restrict__ w,
                                                         float *__restrict__ out, int H, int W, int sin, int sout)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int nb = j / Q, q = j - nb * Q;
    if (nb > W) return;
    const int n = nb - 1, m = (int)blockIdx.y - 1, b = blockIdx.z;
    const int Wo = 2 * W;
    f32x4 v[2][2];            // [input row m+1, m][input column n+1, n]
    bool ok[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int iy = m + 1 - a, ix = n + 1 - c;
            ok[a][c] = iy >= 0 && iy < H && ix >= 0 && ix < W;
            v[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok[a][c]) v[a][c] = *reinterpret_cast<const f32x4 *>(in + ((size_t)(b * H + iy) * W + ix) * sin + 4 * q);
        }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const int oy = 2 * m + 1 + dy;
        if (oy < 0 || oy >= 2 * H) continue;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int ox = 2 * n + 1 + dx;
            if (ox < 0 || ox >= Wo) continue;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (!ok[a][c]) continue;
                    const int ky = dy + 2 * a, kx = dx + 2 * c;      // oy = 2 iy - 1 + ky with iy = m + 1 - a
                    const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + ((ky * 4 + kx) * Q + q) * 4);
                    acc[0] = fmaf(v[a][c][0], wv[0], acc[0]);
                    acc[1] = fmaf(v[a][c][1], wv[1], acc[1]);
                    acc[2] = fmaf(v[a][c][2], wv[2], acc[2]);
                    acc[3] = fmaf(v[a][c][3], wv[3], acc[3]);
                }
            *reinterpret_cast<f32x4 *>(out + ((size_t)(b * 2 * H + oy) * Wo + ox) * sout + 4 * q) = acc;
        }
    }
}

int launch_touch(const float *p, size_t floats, float *sink, hipStream_t st)
{
    const size_t n = floats / 4;
    if (!n) return PIVLFN_OK;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(touch_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const f32x4 *>(p), n, sink);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

int launch_dwconvT(const float *in, const float *w, float *out, int B, int H, int W, int C, int stride_in,
                   int stride_out, int cstore, hipStream_t st)
{
    (void)C;
    const int Q = cstore / 4;
    PIV_REQUIRE(2 * H <= 65535 && B <= 65535, "dwconvT: %d output rows / %d images exceed the grid range", 2 * H, B);
    const dim3 grid((unsigned)cdiv(2 * W * Q, 256), (unsigned)(2 * H), (unsigned)B);
    switch (Q) {
        case 1: hipLaunchKernelGGL(dwconvT_kernel<1>, grid, dim3(256), 0, st, in, w, out, H, W, stride_in, stride_out); break;       // flow (u, v, 0, 0)
        case 14:                                                                                                                     // 49 + 7 correlation lanes
            if (PIV_KNOB(1) & 8388608)      // tools: the one-output-per-thread kernel (A/B)
                hipLaunchKernelGGL(dwconvT_kernel<14>, grid, dim3(256), 0, st, in, w, out, H, W, stride_in, stride_out);
            else
                hipLaunchKernelGGL(dwconvT_b2_kernel<14>, dim3((unsigned)cdiv((W + 1) * Q, 256), (unsigned)(H + 1), (unsigned)B), dim3(256), 0, st, in, w, out, H, W,
                                   stride_in, stride_out);
            break;
        default: PIV_REQUIRE(false, "dwconvT: %d channel quads not instantiated (1 = flow, 14 = correlation)", Q);
    }
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

// ---- Subpixel's backwarp(feat2, flow*scale) (:214): [B,H,W,C] ----------------------------------------------------
__global__ __launch_bounds__(256) void backwarp_nhwc_kernel(const float *__restrict__ in, const float *__restrict__ flow4,
                                                            float scale, float *__restrict__ out, int B, int H, int W, int C)
{
    const unsigned Q = C / 4;
    const unsigned total = (unsigned)B * H * W * Q;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned q = i % Q;
        const unsigned pix = i / Q;
        const int x = (int)(pix % (unsigned)W);
        const unsigned r = pix / (unsigned)W;
        const int y = (int)(r % (unsigned)H);
        const int b = (int)(r / (unsigned)H);
        const float2 uv = *reinterpret_cast<const float2 *>(flow4 + (size_t)pix * 4);
        const Taps t = make_taps((float)x + uv.x * scale, (float)y + uv.y * scale, H, W);
        const float *base = in + (size_t)b * H * W * C + 4 * q;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t.o00 >= 0) v += t.w00 * *reinterpret_cast<const f32x4 *>(base + (size_t)t.o00 * C);
        if (t.o01 >= 0) v += t.w01 * *reinterpret_cast<const f32x4 *>(base + (size_t)t.o01 * C);
        if (t.o10 >= 0) v += t.w10 * *reinterpret_cast<const f32x4 *>(base + (size_t)t.o10 * C);
        if (t.o11 >= 0) v += t.w11 * *reinterpret_cast<const f32x4 *>(base + (size_t)t.o11 * C);
        *reinterpret_cast<f32x4 *>(out + (size_t)pix * C + 4 * q) = v;
    }
}

int launch_backwarp_nhwc(const float *in, const float *flow4, float scale, float *out, int B, int H, int W, int C,
                         hipStream_t st)
{
    PIV_REQUIRE(C % 4 == 0, "backwarp (channels-last): C=%d must be a multiple of 4", C);
    const size_t total = (size_t)B * H * W * (C / 4);
    PIV_REQUIRE(total < 0x7fffffffull, "backwarp: %zu work items exceed the 32-bit index range", total);
    hipLaunchKernelGGL(backwarp_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, st, in, flow4, scale, out, B, H, W, C);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

// ---- per-sample spatial mean of the flow (:275), deterministic two-stage sum ------------------------------------
constexpr int MEAN_BLOCKS = 64;
int flow_mean_partials(int) { return MEAN_BLOCKS; }

__device__ __forceinline__ float2 block_sum2(float2 v, float2 *sh)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        v.x += __shfl_down(v.x, o);
        v.y += __shfl_down(v.y, o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float2 r = make_float2(0.f, 0.f);
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { r.x += sh[w].x; r.y += sh[w].y; }
    return r;
}

__global__ __launch_bounds__(256) void flow_mean_stage1(const float *__restrict__ flow4, float *__restrict__ partial, int HW)
{
    __shared__ float2 sh[4];
    const int b = blockIdx.y;
    float2 s = make_float2(0.f, 0.f);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += MEAN_BLOCKS * 256) {
        const float2 uv = *reinterpret_cast<const float2 *>(flow4 + ((size_t)b * HW + i) * 4);
        s.x += uv.x;
        s.y += uv.y;
    }
    const float2 r = block_sum2(s, sh);
    if (threadIdx.x == 0) {
        partial[((size_t)b * MEAN_BLOCKS + blockIdx.x) * 2 + 0] = r.x;
        partial[((size_t)b * MEAN_BLOCKS + blockIdx.x) * 2 + 1] = r.y;
    }
}

// Second stage of the mean: the MEAN_BLOCKS partial sums of image b, tree over one wave's lanes, / HW.  Every workgroup of
// reg_prep_kernel runs it for itself (64 loads and 6 shuffles: cheaper than the launch a separate one-wave kernel costs in the
// dependent chain of a level); the one-wave kernel stays for the stand-alone entry point.  Same tree, same bits.
__device__ __forceinline__ float2 mean_from_partials(const float *__restrict__ partial, int b, int HW, int lane)
{
    float2 v = make_float2(partial[((size_t)b * MEAN_BLOCKS + lane) * 2], partial[((size_t)b * MEAN_BLOCKS + lane) * 2 + 1]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        v.x += __shfl_down(v.x, o);
        v.y += __shfl_down(v.y, o);
    }
    const float2 r = make_float2(0.f + __shfl(v.x, 0), 0.f + __shfl(v.y, 0));      // block_sum2's final "0 + sh[0]" (exact)
    return make_float2(r.x / (float)HW, r.y / (float)HW);
}

__global__ __launch_bounds__(64) void flow_mean_stage2(const float *__restrict__ partial, float *__restrict__ mean, int HW)
{
    const float2 m = mean_from_partials(partial, blockIdx.x, HW, threadIdx.x);
    if (threadIdx.x == 0) {
        mean[blockIdx.x * 2 + 0] = m.x;
        mean[blockIdx.x * 2 + 1] = m.y;
    }
}

int launch_flow_mean(const float *flow4, float *partial, float *mean, int B, int HW, hipStream_t st)
{
    hipLaunchKernelGGL(flow_mean_stage1, dim3(MEAN_BLOCKS, B), dim3(256), 0, st, flow4, partial, HW);
    if (mean) hipLaunchKernelGGL(flow_mean_stage2, dim3(B), dim3(64), 0, st, partial, mean, HW);      // mean == nullptr: the consumer reduces the partials itself
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

// ---- Regularization front (:275-277): rm = flow - mean; norm = ||img1 - backwarp(img2, flow*scale)||_2 ------------
// grid = (blocks per image, B); `partial` != nullptr: the mean is formed here from flow_mean_stage1's partial sums (and written to
// `mean` by the first workgroup of every image, for callers that read it back); else `mean` is read.
__global__ __launch_bounds__(256) void reg_prep_kernel(const f32x4 *__restrict__ img1, const f32x4 *__restrict__ img2,
                                                       const f32x4 *__restrict__ flow4, float *__restrict__ mean,
                                                       const float *__restrict__ partial,
                                                       float scale, f32x4 *__restrict__ misc4, int H, int W)
{
    const int b = blockIdx.y;
    const unsigned HW = (unsigned)H * W;
    float2 m;
    if (partial) {
        m = mean_from_partials(partial, b, (int)HW, threadIdx.x & 63);        // every wave for itself: no barrier
        if (blockIdx.x == 0 && threadIdx.x == 0) { mean[b * 2 + 0] = m.x; mean[b * 2 + 1] = m.y; }
    } else {
        m = make_float2(mean[b * 2 + 0], mean[b * 2 + 1]);
    }
    const f32x4 *base = img2 + (size_t)b * HW;
    for (unsigned p = blockIdx.x * 256u + threadIdx.x; p < HW; p += gridDim.x * 256u) {
        const int x = (int)(p % (unsigned)W);
        const int y = (int)(p / (unsigned)W);
        const size_t i = (size_t)b * HW + p;
        const f32x4 fl = flow4[i];
        const Taps t = make_taps((float)x + fl[0] * scale, (float)y + fl[1] * scale, H, W);
        f32x4 wv = {0.f, 0.f, 0.f, 0.f};
        if (t.o00 >= 0) wv += t.w00 * base[t.o00];
        if (t.o01 >= 0) wv += t.w01 * base[t.o01];
        if (t.o10 >= 0) wv += t.w10 * base[t.o10];
        if (t.o11 >= 0) wv += t.w11 * base[t.o11];
        const f32x4 d = img1[i] - wv;
        f32x4 o;
        o[0] = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        o[1] = fl[0] - m.x;
        o[2] = fl[1] - m.y;
        o[3] = 0.f;
        misc4[i] = o;
    }
}

int launch_reg_prep(const float *img1, const float *img2, const float *flow4, float *mean, const float *partial, float scale,
                    float *misc4, int B, int H, int W, hipStream_t st)
{
    const size_t per = (size_t)H * W;
    const int gx = (int)std::min<size_t>((per + 255) / 256, (size_t)std::max(1, 8192 / B));
    hipLaunchKernelGGL(reg_prep_kernel, dim3(gx, B), dim3(256), 0, st, reinterpret_cast<const f32x4 *>(img1),
                       reinterpret_cast<const f32x4 *>(img2), reinterpret_cast<const f32x4 *>(flow4), mean, partial, scale,
                       reinterpret_cast<f32x4 *>(misc4), H, W);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

// ---- Regularization tail (:281-302): softmax(-d^2) weighted k x k local average of u and v -----------------------
// e_c = exp(-d_c^2 - max_c(-d_c^2)); Z = 1/sum e; u' = (bx + sum_c wx[c] e_c u[y+ky-p, x+kx-p]) * Z  (bias inside, :288-300)
// One workgroup = a 16 x 16 tile of pixels, one pixel per thread.  The k x k neighbourhood of (u, v) every pixel needs (the
// F.unfold of :290-299) comes from an LDS copy of the tile's (16 + k - 1)^2 flow halo: read from global memory, the 49 neighbour
// loads per pixel were what the 7 x 7 tail at level 1 spent its time on (95 us; the dist rows themselves are 235 MB = 45 us).
template <int K>
__global__ __launch_bounds__(256) void reg_tail_kernel(const float *__restrict__ dist, int dstride,
                                                       const float *__restrict__ flow4, const float *__restrict__ wx,
                                                       const float *__restrict__ wy, float bx, float by,
                                                       float *__restrict__ out4, float *__restrict__ out_nchw,
                                                       float out_scale, int B, int H, int W)
{
    constexpr int KK = K * K, P = K / 2, TW = 16 + K - 1;
    __shared__ float2 fl[TW * TW];
    const size_t img = (size_t)H * W;
    const int tiles_x = (W + 15) >> 4, tiles_y = (H + 15) >> 4;
    int bid = blockIdx.x;
    const int tx0 = (bid % tiles_x) * 16;
    bid /= tiles_x;
    const int ty0 = (bid % tiles_y) * 16;
    const int b = bid / tiles_y;
    const int tid = threadIdx.x;
    for (int i = tid; i < TW * TW; i += 256) {
        const int yy = ty0 + i / TW - P, xx = tx0 + i % TW - P;
        float2 uv = make_float2(0.f, 0.f);                    // zero padding of F.unfold
        if (yy >= 0 && yy < H && xx >= 0 && xx < W)
            uv = *reinterpret_cast<const float2 *>(flow4 + ((size_t)b * img + (size_t)yy * W + xx) * 4);
        fl[i] = uv;
    }
    __syncthreads();
    const int lx = tid & 15, ly = tid >> 4;
    const int x = tx0 + lx, y = ty0 + ly;
    if (x >= W || y >= H) return;
    const size_t i = (size_t)b * img + (size_t)y * W + x;
    float e[KK];
    const float *dp = dist + i * dstride;
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < KK; ++c) {
        const float d = dp[c];
        e[c] = -(d * d);
        m = fmaxf(m, e[c]);
    }
    float z = 0.f, su = 0.f, sv = 0.f;
#pragma unroll
    for (int c = 0; c < KK; ++c) {
        e[c] = expf(e[c] - m);
        z += e[c];
        const float2 uv = fl[(ly + c / K) * TW + lx + c % K];
        su = fmaf(wx[c], e[c] * uv.x, su);
        sv = fmaf(wy[c], e[c] * uv.y, sv);
    }
    const float zi = 1.f / z;
    const float u = (su + bx) * zi, v = (sv + by) * zi;
    if (out4) {
        f32x4 o = {u, v, 0.f, 0.f};
        reinterpret_cast<f32x4 *>(out4)[i] = o;
    }
    if (out_nchw) {
        const size_t pix = (size_t)y * W + x;
        out_nchw[((size_t)b * 2 + 0) * img + pix] = u * out_scale;
        out_nchw[((size_t)b * 2 + 1) * img + pix] = v * out_scale;
    }
}

int launch_reg_tail(const float *dist, int dstride, const float *flow4, const float *wx, const float *wy, float bx,
                    float by, int k, float *out4, float *out_nchw, float out_scale, int B, int H, int W, hipStream_t st)
{
    const dim3 g((unsigned)(cdiv(W, 16) * cdiv(H, 16) * B)), t(256);
    switch (k) {
        case 3: hipLaunchKernelGGL(reg_tail_kernel<3>, g, t, 0, st, dist, dstride, flow4, wx, wy, bx, by, out4, out_nchw, out_scale, B, H, W); break;
        case 5: hipLaunchKernelGGL(reg_tail_kernel<5>, g, t, 0, st, dist, dstride, flow4, wx, wy, bx, by, out4, out_nchw, out_scale, B, H, W); break;
        case 7: hipLaunchKernelGGL(reg_tail_kernel<7>, g, t, 0, st, dist, dstride, flow4, wx, wy, bx, by, out4, out_nchw, out_scale, B, H, W); break;
        default: pivlfn::set_error("reg_tail: k=%d unsupported", k); return PIVLFN_ERR_ARG;
    }
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

// ---- [B,H,W,4] flow -> NCHW [B,2,H,W] (per-level debug output, :363-367) --------------------------------------------
__global__ __launch_bounds__(256) void flow4_to_nchw_kernel(const float *__restrict__ flow4, float *__restrict__ out, int B, int HW)
{
    const size_t total = (size_t)B * HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int b = (int)(i / HW);
        const int pix = (int)(i - (size_t)b * HW);
        const float2 uv = *reinterpret_cast<const float2 *>(flow4 + i * 4);
        out[((size_t)b * 2 + 0) * HW + pix] = uv.x;
        out[((size_t)b * 2 + 1) * HW + pix] = uv.y;
    }
}

int launch_flow4_to_nchw(const float *flow4, float *out, int B, int H, int W, hipStream_t st)
{
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(flow4_to_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, st, flow4, out, B, H * W);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

}  // namespace pivlfn
