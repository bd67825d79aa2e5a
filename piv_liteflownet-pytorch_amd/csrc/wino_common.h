// Shared pieces of the Winograd F(2x2, 3x3) kernels (conv_wino.hip: one workgroup = four waves that each transform and multiply;
// conv_wino_ws.hip: persistent workgroups with transform-producer waves and matrix-only consumer waves).  Both stage the raw
// 8-channel patch in the same de-interleaved LDS image, read the same packed weights (pack_conv_w) and keep the same summation order.
#pragma once
#include "common.h"

namespace pivlfn {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// Differences as fused multiply-adds with -1: the compiler then emits one v_pk_fma_f32 per channel pair (a plain a - b becomes two
// v_sub_f32: it does not fold the negation into v_pk_add_f32's modifiers).  The product is exact, so the value is the fp32
// difference.  `m1` is -1 laundered through an empty asm so that the multiplication is not folded back into a subtraction.  (An
// earlier version used inline-assembly v_pk_add_f32 / v_pk_fma_f32: instructions the compiler's hazard recognizer cannot see --
// nothing waits for an in-flight MFMA result or spaces a matrix instruction behind them -- and two builds that placed them next to
// MFMAs returned wrong values at full occupancy.  Everything here is compiler-generated again.)
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f32x4 sub4(f32x4 a, f32x4 b, float m1)
{
    // two <2 x float> fused multiply-adds: the form the compiler keeps as v_pk_fma_f32 (four scalar fmaf calls are only partly re-packed)
    const f32x2 m = {m1, m1};
    const f32x2 lo = __builtin_elementwise_fma(m, __builtin_shufflevector(b, b, 0, 1), __builtin_shufflevector(a, a, 0, 1));
    const f32x2 hi = __builtin_elementwise_fma(m, __builtin_shufflevector(b, b, 2, 3), __builtin_shufflevector(a, a, 2, 3));
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}
constexpr int WPW = 18;      // patch width in pixels: 8 tiles x 2 + 2
// LDS image of one chunk's patch.  Every operand read is one 16-byte quad per lane at pixel (2 ty + r, 2 tx + c) of the lane's
// tile (ty, tx): a stride of two pixels in both directions, which in a plain row-major image leaves only even 16-byte slots
// (at least a two-way bank conflict; measured 65 % of all LDS cycles with a 32-byte pixel pitch).  The image is therefore stored
// de-interleaved -- even columns then odd columns inside a row, even rows then odd rows -- so tiles are unit-stride in both
// directions, with a pixel pitch of 3 quads (odd) and a row pitch of 8 quads (mod 16): the 16 lanes of every ds_read_b128 group
// ({0-3,12-15,20-27}, ... = tile rows {0,3} x columns 0-3 and rows {1,2} x columns 4-7, or the complement) land on 16 distinct slots.
constexpr int WPIXQ = 3;                       // 16-byte quads per staged pixel: 8 channels of the K chunk + 4 floats of padding
constexpr int WROWQ = 56;                      // quads per patch row: 18 x 3 = 54, padded to 896 bytes = 8 quads mod 16
// (all LDS offsets are kept in quads and applied to an f32x4 pointer, so every access is provably 16-byte aligned: with float
// offsets the compiler splits the 16-byte reads and writes into ds_read2_b32 / ds_write2_b32 pairs)
constexpr unsigned WOOB = 0x80000000u;


}  // namespace pivlfn
