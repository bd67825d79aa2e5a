/* Test driver (not product, not oracle): runs the oracle's C restatement (oracle/corr_oracle.c) under AddressSanitizer and
 * UndefinedBehaviorSanitizer on the edge shapes the GPU tests use -- 1 x 1 maps, odd sizes, strides 1 and 2, channel counts that
 * are not multiples of 4, flows that leave the image -- with exactly-sized heap buffers, so any out-of-bounds index of the
 * checker itself aborts the run.  Built and run by tests/test_oracle_sanitizers.py:  gcc -fsanitize=address,undefined. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

int corr_forward(const float *first, const float *second, float *out, int B, int C, int H, int W, int s);
int backwarp_forward(const float *in, const float *flow, float *out, int B, int C, int H, int W);
int corr_backward(const float *first, const float *second, const float *gout, float *gfirst, float *gsecond, int B, int C, int H, int W, int s);

static unsigned long long rng = 88172645463325252ull;
static float rnd(void)
{
    rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
    return (float)((rng >> 11) & 0xffffff) / 8388608.0f - 1.0f;
}
static float *randbuf(size_t n, float scale)
{
    float *p = (float *)malloc(n * sizeof(float));
    if (!p) exit(3);
    for (size_t i = 0; i < n; ++i) p[i] = scale * rnd();
    return p;
}

int main(void)
{
    static const int shapes[][5] = {   /* B, C, H, W, stride */
        {1, 1, 1, 1, 1}, {1, 3, 1, 1, 2}, {2, 20, 5, 7, 1}, {1, 33, 9, 4, 2}, {1, 64, 16, 16, 2}, {3, 4, 2, 13, 1}, {1, 2, 7, 7, 2},
    };
    double checksum = 0.0;
    for (size_t k = 0; k < sizeof(shapes) / sizeof(shapes[0]); ++k) {
        const int B = shapes[k][0], C = shapes[k][1], H = shapes[k][2], W = shapes[k][3], s = shapes[k][4];
        const int Ho = (H + s - 1) / s, Wo = (W + s - 1) / s;
        const size_t nin = (size_t)B * C * H * W, nout = (size_t)B * 49 * Ho * Wo;
        float *f1 = randbuf(nin, 1.f), *f2 = randbuf(nin, 1.f), *flow = randbuf((size_t)B * 2 * H * W, 3.f * (float)(H > W ? H : W));
        float *out = (float *)malloc(nout * sizeof(float)), *warped = (float *)malloc(nin * sizeof(float));
        float *gout = randbuf(nout, 1.f), *g1 = (float *)calloc(nin, sizeof(float)), *g2 = (float *)calloc(nin, sizeof(float));
        if (!out || !warped || !g1 || !g2) return 3;
        if (corr_forward(f1, f2, out, B, C, H, W, s)) return 4;
        if (backwarp_forward(f2, flow, warped, B, C, H, W)) return 5;     /* flows up to 3 image sizes: every tap outside */
        if (corr_backward(f1, f2, gout, g1, g2, B, C, H, W, s)) return 6;
        if (corr_backward(f1, f2, gout, NULL, g2, B, C, H, W, s)) return 7;
        if (corr_backward(f1, f2, gout, g1, NULL, B, C, H, W, s)) return 8;
        for (size_t i = 0; i < nout; ++i) checksum += out[i];
        for (size_t i = 0; i < nin; ++i) checksum += warped[i] + g1[i] + g2[i];
        free(f1); free(f2); free(flow); free(out); free(warped); free(gout); free(g1); free(g2);
    }
    if (corr_forward(NULL, NULL, NULL, 0, 1, 1, 1, 1) == 0) return 9;      /* empty shapes are refused, not dereferenced */
    if (!isfinite(checksum)) return 10;
    printf("oracle C under ASan + UBSan: %zu shapes, checksum %.6f\n", sizeof(shapes) / sizeof(shapes[0]), checksum);
    return 0;
}
