// Shared by the split-half ("f16x3") matrix-core kernels: kernels_conv_f16x3.hip (conv3x3 C -> C) and kernels_pix2x2_f16x3.hip
// (DRUNet's 2 x 2 stride-2 and transposed convolutions).  DESIGN.md 4.8.
#pragma once
#include "conv_common.h"

namespace pnp {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr float H3_SCALE = 2048.f, H3_RSCALE = 1.f / 2048.f;
constexpr int H3_STR = 68;                       // floats between the staging rows of the epilogue
constexpr int H3_TAP16 = 1024;                   // 16-byte units of one 64 x 64 block of weights: [K step 2][N tile 4][hi, lo][lane 64]

// x = hi + lo / 2048,  hi = half(x),  lo = half((x - hi) * 2048)
__device__ __forceinline__ void split4(const f32x4& v, h4& hi, h4& lo) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        // two values at a time: ONE packed conversion for the hi halves; (x - h) * 2048 as fma(h, -2048, x * 2048) on the half as it is
        // (v_fma_mixlo / mixhi_f16) -- both forms are exact: the residual has at most 13 significant bits
        const f32x2 x = {v[2 * k], v[2 * k + 1]};
        const h2 h = __builtin_convertvector(x, h2);
        hi[2 * k] = h[0]; hi[2 * k + 1] = h[1];
        lo[2 * k] = (_Float16)__builtin_fmaf((float)h[0], -H3_SCALE, x[0] * H3_SCALE);
        lo[2 * k + 1] = (_Float16)__builtin_fmaf((float)h[1], -H3_SCALE, x[1] * H3_SCALE);
    }
}

// The operand tile in LDS (kernels_conv_f16x3.hip): a pixel is 256 bytes of halves + 16 bytes of padding (CV_PS floats), in sixteen
// 16-byte chunks.  Chunk (kb, s2, part) = [hi | lo] halves of input channels 32 s2 + 8 kb .. + 7 -- lane (i, kb) of a wave reads it as the
// A fragment of K step s2 -- sits at h3_chunk_pos: the chunks of kb and kb ^ 1 lie 128 bytes apart, and the wave's M-tile row i is tile
// column h3_row_pixel(i) (columns 0-3 and 4-7 exchanged).  Why: ds_read_b128 serves the lanes in the groups {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS table), i.e. rows {0-3, 12-15} of kb = 0 together with rows 4-11 of kb = 1.  In the
// straightforward order (chunk = 4 s2 + kb, row i = column i: rounds 4's layout) those sixteen 16-byte accesses fall on 15 distinct
// bank quads -- every A read takes 8 LDS cycles instead of 4, a quarter of the kernel's LDS cycles (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
// = 0.23).  With the pair 128 bytes apart and the two column quads exchanged the sixteen accesses of every group cover the 64 banks
// exactly once -- same bytes of LDS, the same immediate offsets per tap, no address arithmetic added.
__device__ __forceinline__ constexpr int h3_chunk_pos(int kb, int s2, int part) { return ((kb & 1) * 8 + (kb >> 1) * 4 + s2 * 2 + part) * 16; }
__device__ __forceinline__ int h3_row_pixel(int i) { return (i & 8) ? i : (i ^ 4); }

// The SPLIT activation format: what the f16x3 kernels hand to each other between layers instead of float32.  Same shape, same bytes --
// [n][H][W][C] with 256 bytes per pixel and block of 64 channels -- but a block is [64 hi halves][64 lo halves] of its 64 values
// (the operand layout of the kernels' LDS tiles): the producing layer splits each output value ONCE in its epilogue, and a consuming
// layer copies 16-byte runs of halves into its tile instead of splitting every value of every tile (halo included) again for every
// block of output channels.  A value read back is hi + lo / 2048: the float32 result rounded to 2^-22 relative -- the precision the
// consumer's operands had anyway.
__device__ __forceinline__ float unsplit(const _Float16 hi, const _Float16 lo) { return fmaf((float)lo, H3_RSCALE, (float)hi); }

}  // namespace pnp
