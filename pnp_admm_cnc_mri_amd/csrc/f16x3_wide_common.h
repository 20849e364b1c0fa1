// Shared by the two wide split-half conv3x3 kernels (kernels_conv_f16x3_wide.hip: any C, kernels_conv_f16x3_pipe.hip: C = 64, epilogue inside the
// tap loop): tile geometry, the helper waves' input staging, the permuted weight-row read.  DESIGN.md 4.8.
#pragma once
#include "conv_common.h"
#include "f16x3_common.h"

namespace pnp {

constexpr int WT_TX = 16, WT_TY = 16;                            // output tile
constexpr int WT_HX = WT_TX + 2, WT_HY = WT_TY + 2;              // with halo
constexpr int WT_PSB = CV_PS * 4;                                // bytes between consecutive pixels of the LDS tile (272)
constexpr int WT_XIN = WT_HY * WT_HX * CV_PS;                    // floats of the input tile (88 128 bytes)
constexpr int WT_HTHREADS = 256;                                 // threads of the four HELPER waves (4..7)
constexpr int WT_XU = (WT_HY * WT_HX * 16 + WT_HTHREADS - 1) / WT_HTHREADS;     // 16-byte chunks of the tile per helper thread: 21
constexpr int WT_THREADS = 512;
constexpr int WT_NBUF = 4;                                       // weight buffers: the stream runs three taps ahead of the MFMAs (buffer of global tap T = T % 4)
constexpr int WT_PIECE = (WT_XU + 5) / 6;                        // the next input tile is requested in six pieces, behind the DMAs of taps 0..5

struct WTilePos { int img, y0, x0; };
__device__ __forceinline__ WTilePos wtile_pos(const ConvArgs& a, int t) {
    const int per_img = a.tiles_x * a.tiles_y;
    WTilePos q;
    q.img = t / per_img;
    const int trem = t - q.img * per_img, ty = trem / a.tiles_x;
    q.y0 = ty * WT_TY; q.x0 = (trem - ty * a.tiles_x) * WT_TX;
    return q;
}

// ---- helper waves: input staging (the narrow kernel's scheme, kernels_conv_f16x3.hip, on the 18 x 18 tile and 256 threads) ----
// chunk u of helper thread h (0..255) is tile pixel p = wstage_pixel(h, u) = (row r, column c), channels 4 (h & 15) ..  Sixteen 16-lane groups:
// the two groups of a 32-lane half take pixels 8 apart (8 x 272 bytes = 32 banks: their 8-byte LDS writes do not collide).
__device__ __forceinline__ constexpr int wstage_pixel(int h, int u) { return ((h >> 4) >> 1) + 8 * ((h >> 4) & 1) + 16 * u; }
// One packed coordinate register per chunk, computed ONCE (the narrow kernel's StagingP): pk[u] = (r W + c) * pix | c -- pix is a multiple of 256,
// the low byte holds the column, the only coordinate that needs a test per tile (rows fall out of the buffer range by themselves) -- or -1: no
// such chunk.  A helper wave's vector instructions share the SIMD's issue port with its partner's MFMAs: fewer of them per load is time for those.
struct WStaging { int pk[WT_XU]; };
__device__ __forceinline__ void wstaging_init(const ConvArgs& a, int h, WStaging& st, const int pix) {
#pragma unroll
    for (int u = 0; u < WT_XU; ++u) {
        const int p = wstage_pixel(h, u), r = p / WT_HX, c = p - r * WT_HX;
        st.pk[u] = (p < WT_HY * WT_HX) ? (((r * a.W + c) * pix) | c) : -1;
    }
}
struct WFetch { __amdgpu_buffer_rsrc_t rs; int origin, xlo, xhi; };
__device__ __forceinline__ WFetch wfetch_begin(const ConvArgs& a, const WTilePos& q, int h, const int pix, const int coff, const bool any = true) {
    WFetch f;
    // any = false: a descriptor of zero bytes -- every piece is still ISSUED (the counted waits below assume it) but none reaches memory
    f.rs = image_rsrc(a.x + (size_t)q.img * a.H * a.W * (pix >> 2), any ? a.H : 0, a.W, pix, any ? coff : 0);
    f.origin = ((q.y0 - 1) * a.W + (q.x0 - 1)) * pix + 16 * (h & 15);        // may be negative: such offsets are out of range as unsigned
    f.xlo = 1 - q.x0; f.xhi = a.W + 1 - q.x0;                                // valid tile columns: xlo <= c < xhi
    return f;
}
template <int U0, int U1>
__device__ __forceinline__ void wfetch_piece(const WFetch& f, const WStaging& st, f32x4 (&v)[WT_XU]) {
#pragma unroll
    for (int u = U0; u < U1 && u < WT_XU; ++u) {
        const int c = st.pk[u] & 255;
        const bool in = st.pk[u] >= 0 && c >= f.xlo && c < f.xhi;
        const int off = in ? f.origin + (st.pk[u] & ~255) : -16;
        const u32x4v w = __builtin_amdgcn_raw_buffer_load_b128(f.rs, off, 0, 0);
        v[u] = f32x4{__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w)};
    }
}
// registers -> LDS tile (split on the way unless the tensor is in the SPLIT activation format already): put_input_h3's layout
template <int U0 = 0, int U1 = WT_XU>
__device__ __forceinline__ void wput_input(float* xin, int h, const f32x4 (&v)[WT_XU], const bool in_split) {
    if (in_split) {
        const int q = h & 15;
        char* px = reinterpret_cast<char*>(xin) + wstage_pixel(h, 0) * WT_PSB + h3_chunk_pos(q & 3, (q >> 2) & 1, q >> 3);
#pragma unroll
        for (int u = U0; u < U1; ++u)
            if (wstage_pixel(h, u) < WT_HY * WT_HX)
                *reinterpret_cast<f32x4*>(px + (wstage_pixel(0, u)) * WT_PSB) = v[u];
        return;
    }
    const int t = h & 15;
    char* px = reinterpret_cast<char*>(xin) + wstage_pixel(h, 0) * WT_PSB + h3_chunk_pos((t >> 1) & 3, t >> 3, 0) + 8 * (t & 1);
#pragma unroll
    for (int u = U0; u < U1; ++u)
        if (wstage_pixel(h, u) < WT_HY * WT_HX) {
            h4 hi, lo;
            split4(v[u], hi, lo);
            *reinterpret_cast<h4*>(px + (wstage_pixel(0, u)) * WT_PSB) = hi;
            *reinterpret_cast<h4*>(px + (wstage_pixel(0, u)) * WT_PSB + 16) = lo;
        }
}

// Which output channel a row of the transposed product is.  The weight fragments reach the compute waves PERMUTED: row m of channel tile
// ct = 2 g + e is output channel 32 g + 8 (m >> 2) + 4 e + (m & 3) of the block.  wt_wrow_offset: where, among the packed fragments of a tap,
// lane (i = m, kb) finds it (relative to fragment nt = 2 g, + 64 e) -- the helper waves' LDS-DMA uses it as its per-lane global address, so the
// permutation costs nothing and the LDS image is read at lane * 16.  An accumulator quad of lane (i, kb) in tile (ct, pt) is then channels
// 32 g + 8 kb + 4 e .. + 3 of pixel (tile row 4 w + pt, column h3_row_pixel(i)): the two tiles of a pair give a lane EIGHT consecutive channels
// -- 32 bytes of a float32 pixel, or 16 bytes of hi halves + 16 bytes of lo halves of a split one -- and every store is 16 bytes straight from
// the registers with no exchange between lanes.
__device__ __forceinline__ int wt_wrow_offset(int i, int kb) { return (i >> 3) * 2048 + (8 * ((i >> 2) & 1) + (i & 3)) * 16 + kb * 256; }

#define WT_WAIT_VM(n_) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n_) : "memory")

}  // namespace pnp
