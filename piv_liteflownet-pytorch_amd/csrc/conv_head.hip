// Flow-head convolution: 32 -> 2 channels, k x k (k = 3, 5, 7), stride 1, bias + residual, no activation
// (reference: the last torch.nn.Conv2d of conv_M / conv_S, /root/reference/src/models.py:161-162, 205-206, and the
// "+ xflow" of :186 / :216).  On the matrix cores this layer wastes 30 of 32 output columns, so it runs on the VALU:
//   * one workgroup = 16 x 16 output pixels, one pixel per lane; the (16+k-1)^2 x 32-channel input patch is staged
//     once in LDS as 128-byte pixel vectors whose 16-byte quads are XOR-swizzled by (column>>1)&7, which makes every
//     ds_read_b128 of a 16-pixel row segment conflict-free for every tap;
//   * the 2 x 32 x k x k weights are read through the scalar cache (wave-uniform addresses -> s_load), so each FMA
//     takes its weight from an SGPR and one 16-byte LDS read feeds 8 FMAs (balanced LDS : VALU on CDNA4);
//   * output is the network's 4-lane flow layout (u, v, 0, 0).
#include "common.h"

namespace pivlfn {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// WLDS: weights staged in LDS and read as broadcast ds_reads instead of through the scalar cache.  On a grid of a few
// workgroups (the coarse levels) the K*K*8 dependent s_load round trips are the whole kernel (22 us for 32x32 pixels); with many
// waves per CU they hide behind each other and the scalar path is the better one (it keeps LDS bandwidth for the patch).
template <int K, bool WLDS>
__global__ __launch_bounds__(256) void conv_head_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                        float b0, float b1, const float *__restrict__ res4,
                                                        float *__restrict__ out4, int B, int H, int W)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int P = K / 2, PW = 16 + K - 1, NPIX = PW * PW;
    const int tiles_x = (W + 15) >> 4, tiles_y = (H + 15) >> 4;
    int bid = xcd_remap(blockIdx.x, tiles_x * tiles_y * B);
    const int tx0 = (bid % tiles_x) * 16;
    bid /= tiles_x;
    const int ty0 = (bid % tiles_y) * 16;
    const int b = bid / tiles_y;
    const int tid = threadIdx.x;
    const float *xb = x + (size_t)b * H * W * 32;
    float *wl = smem + NPIX * 32;
    if (WLDS)
        for (int i = tid; i < K * K * 64; i += 256) wl[i] = w[i];

    for (int idx = tid; idx < NPIX * 8; idx += 256) {
        const int pix = idx >> 3, q = idx & 7;
        const int r = pix / PW, c = pix - r * PW;
        const int iy = ty0 + r - P, ix = tx0 + c - P;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const f32x4 *>(xb + ((size_t)iy * W + ix) * 32 + 4 * q);
        *reinterpret_cast<f32x4 *>(smem + pix * 32 + 4 * (q ^ ((c >> 1) & 7))) = v;
    }
    __syncthreads();

    const int lx = tid & 15, ly = tid >> 4;
    const char *sb = reinterpret_cast<const char *>(smem);
    float a0 = 0.f, a1 = 0.f;
#pragma unroll 1
    for (int ky = 0; ky < K; ++ky) {
        const float *wr = (WLDS ? wl : w) + ky * K * 64;      // [kx][quad][out][4]
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const int c = lx + kx;
            const unsigned base = (unsigned)((ly + ky) * PW + c) * 128u | (unsigned)(((c >> 1) & 7) << 4);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(sb + (base ^ (unsigned)(16 * q)));
                const float *wq = wr + (kx * 8 + q) * 8;
                a0 = fmaf(v[0], wq[0], a0); a0 = fmaf(v[1], wq[1], a0); a0 = fmaf(v[2], wq[2], a0); a0 = fmaf(v[3], wq[3], a0);
                a1 = fmaf(v[0], wq[4], a1); a1 = fmaf(v[1], wq[5], a1); a1 = fmaf(v[2], wq[6], a1); a1 = fmaf(v[3], wq[7], a1);
            }
        }
    }
    const int oy = ty0 + ly, ox = tx0 + lx;
    if (oy < H && ox < W) {
        const size_t pix = ((size_t)b * H + oy) * W + ox;
        f32x4 o = {a0 + b0, a1 + b1, 0.f, 0.f};
        if (res4) {
            const float2 r = *reinterpret_cast<const float2 *>(res4 + pix * 4);
            o[0] += r.x;
            o[1] += r.y;
        }
        *reinterpret_cast<f32x4 *>(out4 + pix * 4) = o;
    }
}

template <int K>
static int launch_head_t(const float *x, const float *w, float b0, float b1, const float *res4, float *out4, int B, int H, int W,
                         hipStream_t st)
{
    constexpr int PW = 16 + K - 1;
    const size_t lds = ((size_t)PW * PW * 32 + K * K * 64) * sizeof(float);
    static LdsAttr attr_a, attr_b;
    if (lds > 64 * 1024) {
        if (int rc = ensure_dyn_lds(attr_a, reinterpret_cast<const void *>(conv_head_kernel<K, false>), (int)lds)) return rc;
        if (int rc = ensure_dyn_lds(attr_b, reinterpret_cast<const void *>(conv_head_kernel<K, true>), (int)lds)) return rc;
    }
    const int nblk = cdiv(W, 16) * cdiv(H, 16) * B;
    // (per image, so that a pair's flow never depends on its batch mates -- the two variants differ in nothing but the weight path,
    //  and produce the same bits, but the rule costs nothing)
    if (cdiv(W, 16) * cdiv(H, 16) <= 512 && !(PIV_KNOB(1) & 1024))
        hipLaunchKernelGGL((conv_head_kernel<K, true>), dim3(nblk), dim3(256), lds, st, x, w, b0, b1, res4, out4, B, H, W);
    else
        hipLaunchKernelGGL((conv_head_kernel<K, false>), dim3(nblk), dim3(256), lds, st, x, w, b0, b1, res4, out4, B, H, W);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

// w: device, [k*k][8][2][4] = (tap, channel quad, output, channel-in-quad)
int launch_conv_head(const float *x, const float *w, float b0, float b1, const float *res4, float *out4, int B, int H, int W,
                     int k, hipStream_t st)
{
    PIV_REQUIRE(x && w && out4 && B > 0 && H > 0 && W > 0, "conv_head: bad arguments");
    switch (k) {
        case 3: return launch_head_t<3>(x, w, b0, b1, res4, out4, B, H, W, st);
        case 5: return launch_head_t<5>(x, w, b0, b1, res4, out4, B, H, W, st);
        case 7: return launch_head_t<7>(x, w, b0, b1, res4, out4, B, H, W, st);
    }
    set_error("conv_head: k=%d unsupported", k);
    return PIVLFN_ERR_ARG;
}

}  // namespace pivlfn
