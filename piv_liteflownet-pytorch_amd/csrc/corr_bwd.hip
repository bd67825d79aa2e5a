// Correlation backward: gradFirst / gradSecond of the 7x7 local correlation, NCHW fp32.
// Replaces kernel_Correlation_updateGradFirst / updateGradSecond and their per-sample launch loops
// (/root/reference/src/correlation.py:106-234, 348-405) -- without the two zero-padded NHWC scratch copies ("rbot0/1") the
// reference keeps from the forward, and in one launch per gradient for the whole batch.
//
// On the stride-s sampling grid (Y, X) in [0,Ho) x [0,Wo), with t = 7(dy+3)+(dx+3):
//   gradFirst [b,c,sY,sX] = 1/C * sum_t gout[b,t,Y,X]       * second[b,c,s(Y+dy),s(X+dx)]
//   gradSecond[b,c,sY,sX] = 1/C * sum_t gout[b,t,Y-dy,X-dx] * first [b,c,s(Y-dy),s(X-dx)]
// (terms whose grid position falls outside are zero); every position off the grid gets an exact 0, as in the reference, whose
// kernels write every element of the gradient (xmin > xmax there when the position is not a multiple of the stride).
// The 49 products are accumulated with fmaf in ascending t, the order of the reference's `for p / for o` loops.
//
// One workgroup = a 16x16 patch of the grid for a group of channels: each thread keeps its 49 gout values in registers for the
// whole channel loop (gout is read once per channel group, not once per channel), the other operand's 22x22 halo patch goes
// through LDS once per channel.  HBM-bound: per channel one read of the operand patch and one write of the gradient.
#include "common.h"

namespace pivlfn {

template <bool SECOND>
__global__ __launch_bounds__(256) void corr_bwd_kernel(const float *__restrict__ other, const float *__restrict__ gout,
                                                       float *__restrict__ grad, int C, int H, int W, int Ho, int Wo, int s,
                                                       int cgroup, int tiles_x)
{
    __shared__ float tile[22][23];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int x0 = (blockIdx.x % tiles_x) * 16, y0 = (blockIdx.x / tiles_x) * 16;
    const int b = blockIdx.z, c_lo = blockIdx.y * cgroup;
    const int c_hi = min(C, c_lo + cgroup);
    const int X = x0 + tx, Y = y0 + ty;
    const bool mine = X < Wo && Y < Ho;

    float g[49];
#pragma unroll
    for (int t = 0; t < 49; ++t) {
        const int dy = t / 7 - 3, dx = t % 7 - 3;
        const int yy = SECOND ? Y - dy : Y, xx = SECOND ? X - dx : X;
        const bool ok = mine && yy >= 0 && yy < Ho && xx >= 0 && xx < Wo;
        g[t] = ok ? gout[(((size_t)b * 49 + t) * Ho + yy) * Wo + xx] : 0.f;
    }
    const float fC = (float)C;
    for (int c = c_lo; c < c_hi; ++c) {
        const float *src = other + ((size_t)b * C + c) * H * W;
        __syncthreads();
        for (int idx = tid; idx < 22 * 22; idx += 256) {
            const int i = idx / 22, j = idx - i * 22;
            const int yy = y0 - 3 + i, xx = x0 - 3 + j;
            tile[i][j] = (yy >= 0 && yy < Ho && xx >= 0 && xx < Wo) ? src[(size_t)(s * yy) * W + s * xx] : 0.f;
        }
        __syncthreads();
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 49; ++t) {
            const int dy = t / 7 - 3, dx = t % 7 - 3;
            const float v = SECOND ? tile[ty + 3 - dy][tx + 3 - dx] : tile[ty + 3 + dy][tx + 3 + dx];
            sum = fmaf(g[t], v, sum);
        }
        if (mine) {
            float *dst = grad + ((size_t)b * C + c) * H * W;
            for (int i = 0; i < s; ++i)
                for (int j = 0; j < s; ++j) {
                    const int yy = s * Y + i, xx = s * X + j;
                    if (yy < H && xx < W) dst[(size_t)yy * W + xx] = (i | j) ? 0.f : sum / fC;
                }
        }
    }
}

int launch_corr_bwd(const float *first, const float *second, const float *gout, float *gfirst, float *gsecond,
                    int B, int C, int H, int W, int s, hipStream_t st)
{
    PIV_REQUIRE(first && second && gout, "corr_bwd: null input");
    PIV_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && s >= 1, "corr_bwd: bad shape B=%d C=%d H=%d W=%d stride=%d", B, C, H, W, s);
    PIV_REQUIRE(B <= 65535, "corr_bwd: batch %d exceeds the grid limit", B);
    const int Ho = cdiv(H, s), Wo = cdiv(W, s);
    const int tiles_x = cdiv(Wo, 16), tiles = tiles_x * cdiv(Ho, 16);
    // enough channel groups to fill the chip, but at least 4 channels per group so the 49 gout loads amortise
    int cgroup = 16;
    while (cgroup > 4 && (long)tiles * cdiv(C, cgroup) * B < 2048) cgroup >>= 1;
    const dim3 grid(tiles, cdiv(C, cgroup), B);
    if (gfirst) hipLaunchKernelGGL(corr_bwd_kernel<false>, grid, dim3(256), 0, st, second, gout, gfirst, C, H, W, Ho, Wo, s, cgroup, tiles_x);
    if (gsecond) hipLaunchKernelGGL(corr_bwd_kernel<true>, grid, dim3(256), 0, st, first, gout, gsecond, C, H, W, Ho, Wo, s, cgroup, tiles_x);
    PIV_CHECK_HIP(hipGetLastError());
    return PIVLFN_OK;
}

}  // namespace pivlfn
