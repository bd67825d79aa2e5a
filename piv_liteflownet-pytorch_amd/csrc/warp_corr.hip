// Fused bilinear back-warp + 7x7 local cost volume (+ LeakyReLU) for gfx950.
//
// Replaces, in one launch and with no intermediate tensor:
//   backwarp(feat2, flow*scale)                      /root/reference/src/models.py:20-35, :171
//   kernel_Correlation_rearrange x2 + zero fills     /root/reference/src/correlation.py:9-34, 288-324
//   kernel_Correlation_updateOutput                  /root/reference/src/correlation.py:36-104, 326-337
//   leaky_relu(., 0.1)                               /root/reference/src/models.py:174-184
// out[b, 7(dy+3)+(dx+3), y, x] = (1/C) sum_c f1[b,c,s*y,s*x] * f2w[b,c,s*(y+dy),s*(x+dx)]  (zeros outside),
// f2w[b,c,Y,X] = bilinear(f2[b,c], X + scale*u[b,Y,X], Y + scale*v[b,Y,X])               (zeros outside).
//
// Structure (one workgroup = 256 threads = an 8x8 tile of output pixels):
//   * the warped second feature map is needed only at the (8+6)^2 = 196 stride-s positions around the tile;
//     those vectors are gathered (4 bilinear taps, channel-contiguous 16-byte loads in the NHWC layout),
//     blended and written ONCE into LDS, channels in chunks of CC; at stride 2 three quarters of the warp
//     of the reference never happens;
//   * lane = output pixel (64 per tile), wave = displacement group (d = wave, wave+4, ...): every lane
//     keeps its f1 vector in registers and streams the f2w vectors with ds_read_b128 (pixel pitch CC+4
//     floats -> conflict-free), 13 private accumulators, no cross-lane reduction at all;
//   * results are transposed through LDS so the [B,Ho,Wo,56] store is made of whole 16-byte lanes;
//   * workgroup ids are remapped so every XCD owns a contiguous band of tiles (halo re-reads hit its L2).
#include "common.h"

namespace pivlfn {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int TO = 8;
constexpr int TP = TO + 6;
constexpr int NPOS = TP * TP;
constexpr int OUTC = 56;

struct WcParams {
    const float *f1, *f2, *flow;
    float *out;
    float scale;
    int B, C, H, W, s, Ho, Wo, leaky;
};


template <int CC, bool NHWC>
__global__ __launch_bounds__(256) void warp_corr_kernel(const WcParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PP = CC + 4;          // LDS pixel pitch (floats)
    constexpr int Q = CC / 4;
    float *f2w = smem;                  // [NPOS][PP]
    float *f1t = smem + NPOS * PP;      // [64][PP]

    const int tiles_x = (p.Wo + TO - 1) / TO, tiles_y = (p.Ho + TO - 1) / TO;
    const int nblk = tiles_x * tiles_y * p.B;
    int bid = xcd_remap(blockIdx.x, nblk);
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int ox0 = tx * TO, oy0 = ty * TO;
    const int tid = threadIdx.x, lane = tid & 63, grp = tid >> 6;
    const int ppx = lane & 7, ppy = lane >> 3;
    const size_t img = (size_t)p.H * p.W;

    float acc[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) acc[k] = 0.f;

    for (int c0 = 0; c0 < p.C; c0 += CC) {
        if (c0) __syncthreads();
        if (NHWC) {
            const float *f2b = p.f2 + (size_t)b * img * p.C + c0;
            for (int idx = tid; idx < NPOS * Q; idx += 256) {
                const int pos = idx / Q, q = idx - pos * Q;
                const int iy = (oy0 + pos / TP - 3) * p.s, ix = (ox0 + pos % TP - 3) * p.s;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                    if (p.flow) {
                        const float2 uv = *reinterpret_cast<const float2 *>(p.flow + ((size_t)b * img + (size_t)iy * p.W + ix) * 4);
                        const Taps t = make_taps((float)ix + uv.x * p.scale, (float)iy + uv.y * p.scale, p.H, p.W);
                        if (t.o00 >= 0) v += t.w00 * *reinterpret_cast<const f32x4 *>(f2b + (size_t)t.o00 * p.C + 4 * q);
                        if (t.o01 >= 0) v += t.w01 * *reinterpret_cast<const f32x4 *>(f2b + (size_t)t.o01 * p.C + 4 * q);
                        if (t.o10 >= 0) v += t.w10 * *reinterpret_cast<const f32x4 *>(f2b + (size_t)t.o10 * p.C + 4 * q);
                        if (t.o11 >= 0) v += t.w11 * *reinterpret_cast<const f32x4 *>(f2b + (size_t)t.o11 * p.C + 4 * q);
                    } else {
                        v = *reinterpret_cast<const f32x4 *>(f2b + ((size_t)iy * p.W + ix) * p.C + 4 * q);
                    }
                }
                *reinterpret_cast<f32x4 *>(f2w + pos * PP + 4 * q) = v;
            }
            const float *f1b = p.f1 + (size_t)b * img * p.C + c0;
            for (int idx = tid; idx < 64 * Q; idx += 256) {
                const int pp = idx / Q, q = idx - pp * Q;
                const int oy = oy0 + (pp >> 3), ox = ox0 + (pp & 7);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (oy < p.Ho && ox < p.Wo)
                    v = *reinterpret_cast<const f32x4 *>(f1b + ((size_t)(oy * p.s) * p.W + ox * p.s) * p.C + 4 * q);
                *reinterpret_cast<f32x4 *>(f1t + pp * PP + 4 * q) = v;
            }
        } else {
            const float *f2b = p.f2 + ((size_t)b * p.C + c0) * img;
            for (int idx = tid; idx < NPOS * CC; idx += 256) {
                const int c = idx / NPOS, pos = idx - c * NPOS;
                const int iy = (oy0 + pos / TP - 3) * p.s, ix = (ox0 + pos % TP - 3) * p.s;
                float v = 0.f;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && c0 + c < p.C) {
                    const float *pl = f2b + (size_t)c * img;
                    if (p.flow) {
                        const float u = p.flow[((size_t)b * 2 + 0) * img + (size_t)iy * p.W + ix];
                        const float w = p.flow[((size_t)b * 2 + 1) * img + (size_t)iy * p.W + ix];
                        const Taps t = make_taps((float)ix + u * p.scale, (float)iy + w * p.scale, p.H, p.W);
                        if (t.o00 >= 0) v += t.w00 * pl[t.o00];
                        if (t.o01 >= 0) v += t.w01 * pl[t.o01];
                        if (t.o10 >= 0) v += t.w10 * pl[t.o10];
                        if (t.o11 >= 0) v += t.w11 * pl[t.o11];
                    } else {
                        v = pl[(size_t)iy * p.W + ix];
                    }
                }
                f2w[pos * PP + c] = v;
            }
            const float *f1b = p.f1 + ((size_t)b * p.C + c0) * img;
            for (int idx = tid; idx < 64 * CC; idx += 256) {
                const int c = idx >> 6, pp = idx & 63;
                const int oy = oy0 + (pp >> 3), ox = ox0 + (pp & 7);
                float v = 0.f;
                if (oy < p.Ho && ox < p.Wo && c0 + c < p.C) v = f1b[(size_t)c * img + (size_t)(oy * p.s) * p.W + ox * p.s];
                f1t[pp * PP + c] = v;
            }
        }
        __syncthreads();

        f32x4 a[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) a[q] = *reinterpret_cast<const f32x4 *>(f1t + lane * PP + 4 * q);
#pragma unroll
        for (int k = 0; k < 13; ++k) {
            const int d = grp + 4 * k;
            if (d < 49) {
                const int dy = d / 7, dx = d - dy * 7;
                const float *src = f2w + ((ppy + dy) * TP + ppx + dx) * PP;
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int q = 0; q < Q; q += 2) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4 *>(src + 4 * q);
                    const f32x4 v1 = *reinterpret_cast<const f32x4 *>(src + 4 * q + 4);
                    s0 = fmaf(a[q][0], v0[0], s0); s0 = fmaf(a[q][1], v0[1], s0);
                    s0 = fmaf(a[q][2], v0[2], s0); s0 = fmaf(a[q][3], v0[3], s0);
                    s1 = fmaf(a[q + 1][0], v1[0], s1); s1 = fmaf(a[q + 1][1], v1[1], s1);
                    s1 = fmaf(a[q + 1][2], v1[2], s1); s1 = fmaf(a[q + 1][3], v1[3], s1);
                }
                acc[k] += s0 + s1;
            }
        }
    }

    // transpose through LDS: [64 pixels][56] with exact zeros in lanes 49..55
    __syncthreads();
    float *ost = smem;
    const float cf = (float)p.C;
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        const int d = grp + 4 * k;
        if (d < 49) {
            float v = acc[k] / cf;
            if (p.leaky) v = lrelu01(v);
            ost[lane * OUTC + d] = v;
        }
    }
    if (grp == 0) {
#pragma unroll
        for (int d = 49; d < OUTC; ++d) ost[lane * OUTC + d] = 0.f;
    }
    __syncthreads();
    if (NHWC) {
        for (int idx = tid; idx < 64 * (OUTC / 4); idx += 256) {
            const int pp = idx / (OUTC / 4), q = idx - pp * (OUTC / 4);
            const int oy = oy0 + (pp >> 3), ox = ox0 + (pp & 7);
            if (oy < p.Ho && ox < p.Wo)
                *reinterpret_cast<f32x4 *>(p.out + ((size_t)(b * p.Ho + oy) * p.Wo + ox) * OUTC + 4 * q) =
                    *reinterpret_cast<const f32x4 *>(ost + pp * OUTC + 4 * q);
        }
    } else {
        for (int idx = tid; idx < 49 * 64; idx += 256) {
            const int d = idx >> 6, pp = idx & 63;
            const int oy = oy0 + (pp >> 3), ox = ox0 + (pp & 7);
            if (oy < p.Ho && ox < p.Wo)
                p.out[((size_t)(b * 49 + d) * p.Ho + oy) * p.Wo + ox] = ost[pp * OUTC + d];
        }
    }
}

template <int CC, bool NHWC>
static int launch_wc(const WcParams &p, hipStream_t st)
{
    const size_t lds = (size_t)(NPOS + 64) * (CC + 4) * sizeof(float);
    static bool attr = false;
    if (!attr && lds > 64 * 1024) {
        PIV_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(warp_corr_kernel<CC, NHWC>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    const int nblk = cdiv(p.Wo, TO) * cdiv(p.Ho, TO) * p.B;
    hipLaunchKernelGGL((warp_corr_kernel<CC, NHWC>), dim3(nblk), dim3(256), lds, st, p);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

int launch_warp_corr(const float *f1, const float *f2, const float *flow, float flow_scale, float *out,
                     int B, int C, int H, int W, int stride, int leaky, bool nhwc, hipStream_t st)
{
    PIV_REQUIRE(f1 && f2 && out, "warp_corr: null pointer");
    PIV_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "warp_corr: empty shape B=%d C=%d H=%d W=%d", B, C, H, W);
    PIV_REQUIRE(stride >= 1 && stride <= 4, "warp_corr: stride=%d unsupported", stride);
    WcParams p{f1, f2, flow, out, flow_scale, B, C, H, W, stride, cdiv(H, stride), cdiv(W, stride), leaky};
    if (nhwc) {
        PIV_REQUIRE(C % 32 == 0, "warp_corr (channels-last): C=%d must be a multiple of 32", C);
        if (C % 64 == 0) return launch_wc<64, true>(p, st);
        return launch_wc<32, true>(p, st);
    }
    if (C % 64 == 0) return launch_wc<64, false>(p, st);
    return launch_wc<32, false>(p, st);   // any C: the last chunk is zero-filled past C
}

// ---- stand-alone back-warp, NCHW (src/models.py:20-35) --------------------------------------------------
__global__ __launch_bounds__(256) void backwarp_nchw_kernel(const float *__restrict__ in, const float *__restrict__ flow,
                                                            float *__restrict__ out, int B, int C, int H, int W)
{
    const size_t img = (size_t)H * W;
    const size_t total = (size_t)B * img;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int b = (int)(i / img);
        const int pix = (int)(i - (size_t)b * img);
        const int y = pix / W, x = pix - y * W;
        const float u = flow[((size_t)b * 2 + 0) * img + pix], v = flow[((size_t)b * 2 + 1) * img + pix];
        const Taps t = make_taps((float)x + u, (float)y + v, H, W);
        for (int c = 0; c < C; ++c) {
            const float *pl = in + ((size_t)b * C + c) * img;
            float r = 0.f;
            if (t.o00 >= 0) r += t.w00 * pl[t.o00];
            if (t.o01 >= 0) r += t.w01 * pl[t.o01];
            if (t.o10 >= 0) r += t.w10 * pl[t.o10];
            if (t.o11 >= 0) r += t.w11 * pl[t.o11];
            out[((size_t)b * C + c) * img + pix] = r;
        }
    }
}

int launch_backwarp_nchw(const float *in, const float *flow, float *out, int B, int C, int H, int W, hipStream_t st)
{
    PIV_REQUIRE(in && flow && out, "backwarp: null pointer");
    PIV_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "backwarp: empty shape");
    const size_t total = (size_t)B * H * W;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(backwarp_nchw_kernel, dim3(grid), dim3(256), 0, st, in, flow, out, B, C, H, W);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

}  // namespace pivlfn
