// Micro-benchmark: HBM write rate of the conv epilogue's store pattern vs a fully coalesced one (67 MB, 128 fp32 channels/pixel).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/store_pattern.hip -o gpurun_out/store_pattern && gpurun_out/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// pattern A: what conv_mfma2's epilogue does: lane = (pixel row 0..31, half hh); one instruction writes 32 pixels x 2 x 16 B
__global__ __launch_bounds__(256) void pat_a(float *out, int tiles_x)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 31, hh = lane >> 5;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int W = tiles_x * 32;
    for (int m = 0; m < 2; ++m) {
        const size_t pix = (size_t)(ty * 8 + wave * 2 + m) * W + tx * 32 + row;
        for (int n = 0; n < 4; ++n)
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {1.f * lane, 2.f, 3.f, 4.f * g};
                *reinterpret_cast<f32x4 *>(out + pix * 128 + n * 32 + 8 * g + 4 * hh) = v;
            }
    }
}
// pattern B: one instruction writes 2 pixels x 512 contiguous bytes
__global__ __launch_bounds__(256) void pat_b(float *out, int tiles_x)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int W = tiles_x * 32;
    for (int m = 0; m < 2; ++m)
        for (int i = 0; i < 16; ++i) {
            const size_t pix = (size_t)(ty * 8 + wave * 2 + m) * W + tx * 32 + i * 2 + (lane >> 5);
            f32x4 v = {1.f * lane, 2.f, 3.f, 4.f * i};
            *reinterpret_cast<f32x4 *>(out + pix * 128 + (lane & 31) * 4) = v;
        }
}
// pattern C: 64-byte pieces (two adjacent g per lane pair) -- what a half-transposed epilogue could do
__global__ __launch_bounds__(256) void pat_c(float *out, int tiles_x)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int W = tiles_x * 32;
    for (int m = 0; m < 2; ++m)
        for (int i = 0; i < 16; ++i) {
            // 8 lanes cover 128 B of one pixel; 64 lanes = 8 pixels x 128 B
            const size_t pix = (size_t)(ty * 8 + wave * 2 + m) * W + tx * 32 + (i & 3) * 8 + (lane >> 3);
            f32x4 v = {1.f * lane, 2.f, 3.f, 4.f * i};
            *reinterpret_cast<f32x4 *>(out + pix * 128 + (i >> 2) * 32 + (lane & 7) * 4) = v;
        }
}
int main()
{
    const int H = 1024, W = 1024, tiles_x = W / 32, blocks = tiles_x * (H / 8);
    float *out;
    (void)hipMalloc(&out, (size_t)H * W * 128 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int pat = 0; pat < 3; ++pat) {
        float best = 1e9f;
        for (int it = 0; it < 20; ++it) {
            (void)hipEventRecord(e0);
            if (pat == 0) pat_a<<<blocks, 256>>>(out, tiles_x);
            else if (pat == 1) pat_b<<<blocks, 256>>>(out, tiles_x);
            else pat_c<<<blocks, 256>>>(out, tiles_x);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (it >= 3 && ms < best) best = ms;
        }
        printf("pattern %c: %.1f us  %.2f TB/s\n", 'A' + pat, best * 1e3, (double)H * W * 512 / best / 1e9);
    }
    return 0;
}
