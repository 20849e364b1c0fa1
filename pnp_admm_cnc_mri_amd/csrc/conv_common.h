// Shared by the convolution kernels (kernels_conv.hip: float32 matrix cores; kernels_conv_f16x3.hip: split-half arithmetic):
// tile geometry, the input staging (global -> registers with out-of-image pixels read as zeros) and the epilogue.
#pragma once
#include "internal.h"
#include <hip/hip_runtime.h>

namespace pnp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef CV_MT_N
#define CV_MT_N 1
#endif
constexpr int CV_MT = CV_MT_N;                 // M tiles (of 32 pixels = 2 tile rows) per wave: 1 -> 8 x 16 tiles, two workgroups per unit; 2 -> 16 x 16, one
constexpr int CV_TX = 16, CV_TY = 8 * CV_MT;   // output tile
constexpr int CV_C = 64;                       // channels in and out
constexpr int CV_PS = CV_C + 4;                // floats between consecutive pixels of the LDS tile: 272 bytes, so that the 16 lanes a b128
                                               // read serves per cycle (consecutive pixels, same channels) start 4 banks apart -- and the
                                               // eight operand groups of a tap are IMMEDIATE offsets of one address (no per-group VALU)
constexpr int CV_THREADS = 256;
constexpr int CV_GROUPS = 9 * 8;               // operand groups per tile: 9 taps x 8 groups of 8 input channels
constexpr int CV_WFRAG = CV_GROUPS * 2 * 64 * 4;          // floats of the packed weights (= 9 * 64 * 64)
#ifndef CV_BD_N
#define CV_BD_N 3
#endif
#ifndef CV_WPS
#define CV_WPS 2                               // workgroups per compute unit = waves per SIMD (MT = 1): 2 or 3
#endif
#ifndef CV_RING
#define CV_RING 8                              // register slots of the weight-fragment ring (a power of two > CV_BD_N, dividing 8)
#endif
// Tile geometry by dilation DIL (1: the plain stacks and DRUNet; 2..4: IRCNN's dilated layers, models/network_dncnn.py:87-101): the
// halo is DIL pixels wide, the taps lie DIL pixels apart.  48 / 64 KiB of LDS at DIL 1 / 2 (two workgroups per compute unit), 82 /
// 102 KiB at DIL 3 / 4 (one).
template <int DIL> struct Geo {
    static constexpr int HX = CV_TX + 2 * DIL, HY = CV_TY + 2 * DIL;                    // tile with halo
    static constexpr int XIN = HY * HX * CV_PS;                                         // floats of the input tile
    static constexpr int XU = (HY * HX * 16 + CV_THREADS - 1) / CV_THREADS;             // its 16-byte chunks per thread (12 at DIL 1)
    static constexpr int WPS = (CV_MT == 1 && XIN * 4 <= 80 * 1024) ? CV_WPS : 1;       // workgroups per compute unit
};
constexpr int CV_BD = CV_BD_N;                 // weight fragments are requested this many groups ahead of their MFMAs (ring of CV_RING register slots)

#ifdef CV_PROF
// diagnostic build (profiles/variants.sh build kernels_conv.hip prof "-DCV_PROF"): shader-clock sums per phase, wave 0 of every workgroup
__device__ unsigned long long g_cvprof[1024 * 8];
#define CV_STAMP(k) { const unsigned long long t_ = __builtin_readcyclecounter(); if (tid == 0) psum[k] += t_ - tlast; tlast = t_; }
#else
#define CV_STAMP(k)
#endif

struct ConvArgs {
    const float* x;       // [n][H][W][64]
    const float* w;       // packed: [group G = tap * 8 + g][N tile 2][lane 64][4]  (k_conv_pack_w)
    const float* bias;    // [64] or null
    const float* skip;    // [n][H][W][64] or null: added AFTER bias (and before the ReLU, if any) -- residual blocks
    float* y;             // [n][H][W][64]
    int n, H, W, tiles_x, tiles_y, relu;
    int C;                // kernels_conv_f16x3.hip: channels in = out, a multiple of 64 (the float32 kernel is 64 only)
    int fmt;              // kernels_conv_f16x3.hip: which tensors are in the SPLIT activation format (f16x3_common.h): CV_FMT_X | CV_FMT_SKIP | CV_FMT_Y
};
constexpr int CV_FMT_X = 1, CV_FMT_SKIP = 2, CV_FMT_Y = 4;

// where tile `t` of the launch lies
struct TilePos { int img, y0, x0; };
__device__ __forceinline__ TilePos tile_pos(const ConvArgs& a, int t) {
    const int per_img = a.tiles_x * a.tiles_y;
    TilePos q;
    q.img = t / per_img;
    const int trem = t - q.img * per_img, ty = trem / a.tiles_x;
    q.y0 = ty * CV_TY; q.x0 = (trem - ty * a.tiles_x) * CV_TX;
    return q;
}

// Per-thread constants of the input staging, computed ONCE: vector instructions issued beside the partner wave's MFMA
// stream are slow and cost that stream issue slots (phase clocks, profiles/conv_variants_r04.txt), so everything that does not
// depend on the tile is out of the loop.  Chunk u of thread tid is (tile pixel p = (tid + 256 u) >> 4, channels 4 cq ..).
template <int DIL> struct Staging {
    int goff[Geo<DIL>::XU];     // byte offset of the chunk relative to the tile's first halo pixel (row y0 - DIL, column x0 - DIL), or < 0: none
    int col[Geo<DIL>::XU];      // tile column of the pixel: the only coordinate that needs a test (rows fall out of the buffer range)
    int loff[Geo<DIL>::XU];     // float offset in the LDS tile
};
template <int DIL>
__device__ __forceinline__ void staging_init(const ConvArgs& a, int tid, Staging<DIL>& st, const int pix = CV_C * 4) {
    constexpr int HX = Geo<DIL>::HX, HY = Geo<DIL>::HY;
#pragma unroll
    for (int u = 0; u < Geo<DIL>::XU; ++u) {
        const int idx = tid + CV_THREADS * u, p = idx >> 4, cq = idx & 15, r = p / HX, c = p - r * HX;
        const bool any = idx < HY * HX * 16;
        st.goff[u] = any ? (r * a.W + c) * pix + cq * 16 : -1;
        st.col[u] = c;
        st.loff[u] = any ? p * CV_PS + cq * 4 : 0;
    }
}
// the input tile of `q` (halo included, zeros outside the image): global -> registers.  Buffer loads with ONE 32-bit offset per
// access: rows above / below the image fall outside the descriptor's range by themselves (the offset wraps or exceeds it) and
// the hardware returns zeros; columns left / right of it get such an offset by one select -- no branch around any load.  (As
// `in ? *ptr : zero` hipcc branched around every one of the twelve loads and waited vmcnt(0) behind each: twelve dependent
// memory round trips per tile.)
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
// `pix`: bytes between consecutive pixels of the tensor (4 x its channel count); `coff`: first channel of the 64 this call touches
__device__ __forceinline__ __amdgpu_buffer_rsrc_t image_rsrc(const float* base, int H, int W, const int pix = CV_C * 4, const int coff = 0) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + coff), 0, (int)((unsigned)H * (unsigned)W * (unsigned)pix - 4u * (unsigned)coff), 0x00020000);
}
template <int DIL>
__device__ __forceinline__ void fetch_input(const ConvArgs& a, const TilePos& q, const Staging<DIL>& st, f32x4 (&v)[Geo<DIL>::XU],
                                            const int pix = CV_C * 4, const int coff = 0) {
    const __amdgpu_buffer_rsrc_t rs = image_rsrc(a.x + (size_t)q.img * a.H * a.W * (pix >> 2), a.H, a.W, pix, coff);
    const int origin = ((q.y0 - DIL) * a.W + (q.x0 - DIL)) * pix;        // may be negative: such offsets are out of range as unsigned
    const int xlo = DIL - q.x0, xhi = a.W + DIL - q.x0;                          // valid tile columns: xlo <= c < xhi
#pragma unroll
    for (int u = 0; u < Geo<DIL>::XU; ++u) {
        const bool in = st.goff[u] >= 0 && st.col[u] >= xlo && st.col[u] < xhi;
        const int off = in ? origin + st.goff[u] : -16;
        const u32x4v w = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        v[u] = f32x4{__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w)};
    }
}

int conv_compute_units();       // kernels_conv.hip

// torch.nn.ReLU: NaN in -> NaN out (fmaxf is IEEE maxNum and would return the 0).  A non-finite activation -- an operand of the f16x3
// kernel beyond the half range, say -- must stay visible through every following layer and the solver's clamp (S6:306-308).
__device__ __forceinline__ float relu_keep_nan(float v) { return v < 0.f ? 0.f : v; }

// Second half of a tile's epilogue: the wave's 32 pixels x 64 channels lie in `stage` in pixel order (row m = 16 * tile row + column,
// STR floats apart) and leave as EIGHT 16-byte stores per lane, a whole 256-byte pixel per 16 lanes: y = relu?(staged + skip).
// Pixel `it` of this lane's eight: tile row 2 MT w + 2 mt + (it >> 2), column 4 (it & 3) + (lane >> 4), channels 4 (lane & 15) ..;
// rows below the image are out of the buffer's range (the store is dropped), columns right of it get such an offset.
// bias64 (f16x3 kernel): the 64 biases of this block of output channels, added here -- in the staged order a lane holds four consecutive
// channels, so the bias is ONE 16-byte value that starts the sum the skip input joins anyway.
template <int STR>
__device__ __forceinline__ void store_rows32(const ConvArgs& a, const TilePos& q, const float* stage, int wv, int lane, int mt,
                                             const int pix = CV_C * 4, const int coff = 0, const float* bias64 = nullptr) {
    const __amdgpu_buffer_rsrc_t ry = image_rsrc(a.y + (size_t)q.img * a.H * a.W * (pix >> 2), a.H, a.W, pix, coff);
    const __amdgpu_buffer_rsrc_t rk = image_rsrc((a.skip ? a.skip : a.y) + (size_t)q.img * a.H * a.W * (pix >> 2), a.H, a.W, pix, coff);
    const int l4 = lane >> 4;
    const int obase = ((q.y0 + 2 * CV_MT * wv) * a.W + q.x0 + l4) * pix + (lane & 15) * 16;
    const int wlim = a.W - q.x0 - l4;                            // column 4 (it & 3) valid iff < wlim
    int off[8];
#pragma unroll
    for (int it = 0; it < 8; ++it)
        off[it] = (4 * (it & 3) < wlim) ? obase + ((2 * mt + (it >> 2)) * a.W + 4 * (it & 3)) * pix : -16;
    const f32x4 b4 = bias64 ? *reinterpret_cast<const f32x4*>(bias64 + 4 * (lane & 15)) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 sk[8];
    if (a.skip) {                                                // all eight requests first: one memory round trip, not eight
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const u32x4v k = __builtin_amdgcn_raw_buffer_load_b128(rk, off[it], 0, 0);
            sk[it] = f32x4{__uint_as_float(k.x), __uint_as_float(k.y), __uint_as_float(k.z), __uint_as_float(k.w)};
        }
        if (bias64) {
#pragma unroll
            for (int it = 0; it < 8; ++it) sk[it] += b4;
        }
    } else {
#pragma unroll
        for (int it = 0; it < 8; ++it) sk[it] = b4;
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        f32x4 v = *reinterpret_cast<const f32x4*>(stage + (4 * it + l4) * STR + (lane & 15) * 4) + sk[it];
        if (a.relu) { v[0] = relu_keep_nan(v[0]); v[1] = relu_keep_nan(v[1]); v[2] = relu_keep_nan(v[2]); v[3] = relu_keep_nan(v[3]); }
        const u32x4v o = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
        __builtin_amdgcn_raw_buffer_store_b128(o, ry, off[it], 0, 0);
    }
}

// The epilogue of a tile for 32x32 accumulator tiles: y = relu?(acc + skip) for the wave's 32 x 64 outputs (the bias is already in
// the accumulators).  Accumulator (reg r, lane) = pixel (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the wave's 32, channel lane & 31
// (+ 32 for the second tile).  Stored straight from there every lane would issue 32 dword stores per tile, and such a tail is
// bound by store ISSUE, not by bandwidth: the outputs go through the wave's own 8 KiB of the (now idle) input tile instead.
__device__ __forceinline__ void store_tile(const ConvArgs& a, const TilePos& q, float* stage, int wv, int lane,
                                           const f32x16 (&acc0)[CV_MT], const f32x16 (&acc1)[CV_MT],
                                           const int pix = CV_C * 4, const int coff = 0) {
    const int i = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int mt = 0; mt < CV_MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * kh;
            stage[m * CV_C + i] = acc0[mt][r];
            stage[m * CV_C + i + 32] = acc1[mt][r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // a wave's LDS instructions execute in order: compiler-only ordering
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        store_rows32<CV_C>(a, q, stage, wv, lane, mt, pix, coff);
        if (mt + 1 < CV_MT) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

}  // namespace pnp
