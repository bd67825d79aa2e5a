/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into, imported by or called from the product
 * path (piv_liteflownet-pytorch_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it, and only as the checker / reported baseline.
 *
 * Scalar C restatement of the reference's CuPy CUDA correlation forward and of its bilinear
 * back-warp.  The loops follow the kernel text one for one, including the zero-padded NHWC
 * scratch ("rbot"), the 32-lane strided partial sums and thread 0's serial reduction, so the fp32
 * summation order is the reference's:
 *   corr_rearrange     <- kernel_Correlation_rearrange     /root/reference/src/correlation.py:9-34
 *   corr_update_output <- kernel_Correlation_updateOutput  /root/reference/src/correlation.py:36-104
 *   corr_forward       <- _FunctionCorrelation.forward     /root/reference/src/correlation.py:287-344
 *   backwarp_forward   <- backwarp()                       /root/reference/src/models.py:20-35
 *                         (grid_sample bilinear / zeros / align_corners=True, restated in pixel units)
 *   corr_grad_first    <- kernel_Correlation_updateGradFirst  /root/reference/src/correlation.py:106-166
 *   corr_grad_second   <- kernel_Correlation_updateGradSecond /root/reference/src/correlation.py:168-234
 *   corr_backward      <- _FunctionCorrelation.backward       /root/reference/src/correlation.py:348-405
 *                         (the backward cannot be executed from the reference here -- it exists only as CuPy CUDA text --
 *                          so its pin is: this loop-for-loop restatement == torch autograd through the pinned forward)
 * nvcc contracts `sum += a*b` into an FMA by default, hence fmaf below.
 *
 * Parity pin: checked against the shimmed import of the reference's own src/models.py in this
 * container by oracle/gen_golden.py (see DESIGN.md "Oracle").
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* input [B,C,H,W] -> output [B,H+6s,W+6s,C], interior only (borders stay zero). */
static void corr_rearrange(const float *input, float *output, int B, int C, int H, int W, int s)
{
    const int PH = H + 6 * s, PW = W + 6 * s;
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int idx = 0; idx < H * W; ++idx) {
                float v = input[((size_t)(b * C + c) * H * W) + idx];
                int py = idx / W + 3 * s;
                int px = idx % W + 3 * s;
                int re = PW * py + px;
                output[((size_t)b * PH * PW + re) * C + c] = v;
            }
}

/* one "block" per output pixel, 32 "threads" striding over channels. */
static void corr_update_output(const float *rbot0, const float *rbot1, float *top,
                               int B, int C, int PH, int PW, int Ho, int Wo, int s)
{
    float *patch = (float *)malloc(sizeof(float) * (size_t)C);
    for (int item = 0; item < B; ++item)
        for (int by = 0; by < Ho; ++by)
            for (int bx = 0; bx < Wo; ++bx) {
                int x1 = (bx + 3) * s;
                int y1 = (by + 3) * s;
                for (int ch = 0; ch < C; ++ch)
                    patch[ch] = rbot0[(((size_t)item * PH + y1) * PW + x1) * C + ch];
                for (int tc = 0; tc < 49; ++tc) {
                    float sum[32];
                    int s2o = (tc % 7 - 3) * s;
                    int s2p = (tc / 7 - 3) * s;
                    int x2 = x1 + s2o, y2 = y1 + s2p;
                    for (int lane = 0; lane < 32; ++lane) {
                        float acc = 0.0f;
                        for (int ch = lane; ch < C; ch += 32)
                            acc = fmaf(patch[ch], rbot1[(((size_t)item * PH + y2) * PW + x2) * C + ch], acc);
                        sum[lane] = acc;
                    }
                    float total = 0.0f;
                    for (int i = 0; i < 32; ++i) total += sum[i];
                    top[(((size_t)item * 49 + tc) * Ho + by) * Wo + bx] = total / (float)C;
                }
            }
    free(patch);
}

/* first, second: [B,C,H,W] fp32 contiguous; out: [B,49,ceil(H/s),ceil(W/s)]. returns 0 on success. */
int corr_forward(const float *first, const float *second, float *out, int B, int C, int H, int W, int s)
{
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || s <= 0) return 1;
    const int PH = H + 6 * s, PW = W + 6 * s;
    const int Ho = (H + s - 1) / s, Wo = (W + s - 1) / s;
    size_t n = (size_t)B * PH * PW * C;
    float *rbot0 = (float *)calloc(n, sizeof(float));
    float *rbot1 = (float *)calloc(n, sizeof(float));
    if (!rbot0 || !rbot1) { free(rbot0); free(rbot1); return 2; }
    corr_rearrange(first, rbot0, B, C, H, W, s);
    corr_rearrange(second, rbot1, B, C, H, W, s);
    corr_update_output(rbot0, rbot1, out, B, C, PH, PW, Ho, Wo, s);
    free(rbot0); free(rbot1);
    return 0;
}

/* out[b,c,y,x] = bilinear(in[b,c], x + flow[b,0,y,x], y + flow[b,1,y,x]); zeros outside. */
int backwarp_forward(const float *in, const float *flow, float *out, int B, int C, int H, int W)
{
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 1;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float fx = (float)x + flow[((size_t)(b * 2 + 0) * H + y) * W + x];
                float fy = (float)y + flow[((size_t)(b * 2 + 1) * H + y) * W + x];
                float x0f = floorf(fx), y0f = floorf(fy);
                float ax = fx - x0f, ay = fy - y0f;
                long x0 = (long)x0f, y0 = (long)y0f;
                float w00 = (1.f - ax) * (1.f - ay), w01 = ax * (1.f - ay);
                float w10 = (1.f - ax) * ay, w11 = ax * ay;
                int v00 = (x0 >= 0 && x0 < W && y0 >= 0 && y0 < H);
                int v01 = (x0 + 1 >= 0 && x0 + 1 < W && y0 >= 0 && y0 < H);
                int v10 = (x0 >= 0 && x0 < W && y0 + 1 >= 0 && y0 + 1 < H);
                int v11 = (x0 + 1 >= 0 && x0 + 1 < W && y0 + 1 >= 0 && y0 + 1 < H);
                for (int c = 0; c < C; ++c) {
                    const float *p = in + (size_t)(b * C + c) * H * W;
                    float acc = 0.f;
                    if (v00) acc += w00 * p[y0 * W + x0];
                    if (v01) acc += w01 * p[y0 * W + x0 + 1];
                    if (v10) acc += w10 * p[(y0 + 1) * W + x0];
                    if (v11) acc += w11 * p[(y0 + 1) * W + x0 + 1];
                    out[((size_t)(b * C + c) * H + y) * W + x] = acc;
                }
            }
    return 0;
}


/* ---- backward ------------------------------------------------------------------------ */
#define ROUND_OFF 50000

/* one "thread" per (c, x, y) of gradFirst, sample b; rbot1 = padded NHWC copy of `second`. */
static void corr_grad_first(int b, const float *rbot1, const float *gout, float *gfirst,
                            int C, int H, int W, int Ho, int Wo, int s)
{
    const int PH = H + 6 * s, PW = W + 6 * s;
    const int total = C * H * W;
    for (int idx = 0; idx < total; ++idx) {
        const int n = idx % C;
        const int l = (idx / C) % W + 3 * s;
        const int m = (idx / C / W) % H + 3 * s;
        const int round_off = ROUND_OFF, round_off_s1 = s * round_off;
        int xmin = (l - 3 * s + round_off_s1 - 1) / s + 1 - round_off;
        int ymin = (m - 3 * s + round_off_s1 - 1) / s + 1 - round_off;
        int xmax = (l - 3 * s + round_off_s1) / s - round_off;
        int ymax = (m - 3 * s + round_off_s1) / s - round_off;
        float sum = 0.f;
        if (xmax >= 0 && ymax >= 0 && xmin <= Wo - 1 && ymin <= Ho - 1) {
            xmin = xmin > 0 ? xmin : 0;
            xmax = xmax < Wo - 1 ? xmax : Wo - 1;
            ymin = ymin > 0 ? ymin : 0;
            ymax = ymax < Ho - 1 ? ymax : Ho - 1;
            for (int p = -3; p <= 3; ++p)
                for (int o = -3; o <= 3; ++o) {
                    const int s2o = s * o, s2p = s * p;
                    const float bot1tmp = rbot1[(((size_t)b * PH + (m + s2p)) * PW + (l + s2o)) * C + n];
                    const int op = (p + 3) * 7 + (o + 3);
                    for (int y = ymin; y <= ymax; ++y)
                        for (int x = xmin; x <= xmax; ++x)
                            sum = fmaf(gout[(((size_t)b * 49 + op) * Ho + y) * Wo + x], bot1tmp, sum);
                }
        }
        gfirst[(((size_t)b * C + n) * H + (m - 3 * s)) * W + (l - 3 * s)] = sum / (float)C;
    }
}

/* one "thread" per (c, x, y) of gradSecond, sample b; rbot0 = padded NHWC copy of `first`. */
static void corr_grad_second(int b, const float *rbot0, const float *gout, float *gsecond,
                             int C, int H, int W, int Ho, int Wo, int s)
{
    const int PH = H + 6 * s, PW = W + 6 * s;
    const int total = C * H * W;
    for (int idx = 0; idx < total; ++idx) {
        const int n = idx % C;
        const int l = (idx / C) % W + 3 * s;
        const int m = (idx / C / W) % H + 3 * s;
        const int round_off = ROUND_OFF, round_off_s1 = s * round_off;
        float sum = 0.f;
        for (int p = -3; p <= 3; ++p)
            for (int o = -3; o <= 3; ++o) {
                const int s2o = s * o, s2p = s * p;
                int xmin = (l - 3 * s - s2o + round_off_s1 - 1) / s + 1 - round_off;
                int ymin = (m - 3 * s - s2p + round_off_s1 - 1) / s + 1 - round_off;
                int xmax = (l - 3 * s - s2o + round_off_s1) / s - round_off;
                int ymax = (m - 3 * s - s2p + round_off_s1) / s - round_off;
                if (xmax >= 0 && ymax >= 0 && xmin <= Wo - 1 && ymin <= Ho - 1) {
                    xmin = xmin > 0 ? xmin : 0;
                    xmax = xmax < Wo - 1 ? xmax : Wo - 1;
                    ymin = ymin > 0 ? ymin : 0;
                    ymax = ymax < Ho - 1 ? ymax : Ho - 1;
                    const float bot0tmp = rbot0[(((size_t)b * PH + (m - s2p)) * PW + (l - s2o)) * C + n];
                    const int op = (p + 3) * 7 + (o + 3);
                    for (int y = ymin; y <= ymax; ++y)
                        for (int x = xmin; x <= xmax; ++x)
                            sum = fmaf(gout[(((size_t)b * 49 + op) * Ho + y) * Wo + x], bot0tmp, sum);
                }
            }
        gsecond[(((size_t)b * C + n) * H + (m - 3 * s)) * W + (l - 3 * s)] = sum / (float)C;
    }
}

/* gfirst / gsecond may be NULL (needs_input_grad false). */
int corr_backward(const float *first, const float *second, const float *gout, float *gfirst, float *gsecond,
                  int B, int C, int H, int W, int s)
{
    const int PH = H + 6 * s, PW = W + 6 * s;
    const int Ho = (H + s - 1) / s, Wo = (W + s - 1) / s;
    float *rbot0 = (float *)calloc((size_t)B * PH * PW * C, sizeof(float));
    float *rbot1 = (float *)calloc((size_t)B * PH * PW * C, sizeof(float));
    if (!rbot0 || !rbot1) {
        free(rbot0);
        free(rbot1);
        return 1;
    }
    corr_rearrange(first, rbot0, B, C, H, W, s);
    corr_rearrange(second, rbot1, B, C, H, W, s);
    for (int b = 0; b < B; ++b) {
        if (gfirst) corr_grad_first(b, rbot1, gout, gfirst, C, H, W, Ho, Wo, s);
        if (gsecond) corr_grad_second(b, rbot0, gout, gsecond, C, H, W, Ho, Wo, s);
    }
    free(rbot0);
    free(rbot1);
    return 0;
}
