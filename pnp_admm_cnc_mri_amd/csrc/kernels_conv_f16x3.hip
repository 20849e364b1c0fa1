// The conv3x3 layer of kernels_conv.hip (64 -> 64 channels, dilation 1..4) -- and C -> C channels for C = 128 .. 1024 -- on the
// HALF-precision matrix cores with float32 results: "f16x3".  DESIGN.md 4.8.
//
// gfx950 has no float32 matrix instruction faster than its vector units (v_mfma_f32_32x32x2_f32: 64 FLOP/clk/SIMD), but the f16
// forms run at 1024 FLOP/clk/SIMD with EXACT products and float32 accumulation.  Every float32 operand is split into two halves,
//
//     x = hi + lo / 2048,   hi = half(x),   lo = half((x - hi) * 2048)          (|x - hi - lo / 2048| <= 2^-22 |x|)
//
// and a product a * b becomes three matrix instructions: hi*hi into one accumulator, hi*lo + lo*hi into a second one that is
// scaled by 1 / 2048 and added at the end (lo*lo, 2^-22 of the product, is dropped).  Half products are exact in float32 (11 x
// 11 significant bits), the accumulation is float32 as in the float32 kernel: the layer's distance from the float64 result
// is that of the float32 kernel (tests/test_gpu_conv.py measures both).  The factor 2048 keeps `lo` a normal half for every
// |x| >= 2^-25; operands beyond +-65504 (the half range) turn into infinities -- loudly, not silently.  Three instructions
// of 16 x the rate: 5.3 x the float32 matrix peak on paper; measured 1.9 - 2.6 x, bound by board power.
//
// Structure (the float32 kernel's, conv_common.h, where it still fits): persistent 256-thread workgroups, two per compute
// unit at dilation 1 (80 KiB of LDS each; one at dilations 2..4: 98 .. 137 KiB), 8 x 16 output pixels x 64 output channels per ITEM, the input tile (halo included) of 64 input channels in LDS for nine
// taps -- already split: a pixel is 128 hi + 128 lo halves + 16 bytes in a bank-conflict-free chunk order (f16x3_common.h), so an operand
// fragment (8 consecutive channels of a pixel) is one ds_read_b128.  What differs:
//   instruction   v_mfma_f32_16x16x32_f16 (the 32x32x16 form does the same arithmetic in the same cycles 7 % slower: under the power
//                 limit the narrow form holds the higher clock); a wave = 2 M tiles (its two tile rows) x 4 N tiles, main and
//                 correction accumulators: 64 registers
//   weights       the float32 kernel streams them from L2 into registers, 147 KiB per wave and tile -- at this arithmetic rate 85
//                 bytes per clock and compute unit, more than the vector memory path delivers.  Here a tap's 16 KiB (64 x 64 x
//                 [hi, lo] in fragment order) go through LDS, double-buffered, by LDS-DMA (buffer_load_dwordx4 ... lds, the tap in
//                 the scalar offset) one tap ahead; a counted s_waitcnt vmcnt and ONE raw s_barrier per tap
//   C channels    an item's K loop runs over C / 64 chunks of input channels x 9 taps; a workgroup keeps its block of 64 output
//                 channels, so every XCD's L2 holds the weights of its blocks while the input tiles stream
//   prefetch      the next input tile is requested in six pieces, one behind each of taps 0..5's DMA (in-order return: a tile
//                 requested at once would be waited for at the next weight wait), issued unconditionally so that hipcc counts them
//   first chunk   its own instance of the code: the accumulators start from a constant-zero C operand, not from 64 register moves
#include "conv_common.h"
#include "f16x3_common.h"
#include <type_traits>
#include <cstdlib>

namespace pnp {

// (h8 / h4, H3_SCALE, H3_STR, H3_TAP16, split4: f16x3_common.h, shared with kernels_pix2x2_f16x3.hip)
template <int DIL> struct GeoH {
    static constexpr int LDS = Geo<DIL>::XIN * 4 + 2 * H3_TAP16 * 16;
    static constexpr int WPS = (CV_MT == 1 && LDS <= 80 * 1024) ? 2 : 1;        // 81 728 bytes at dilation 1: two workgroups fill the 160 KiB
};
static_assert(CV_MT == 1, "the f16x3 kernel is written for 8 x 16 pixel tiles (two tile rows per wave)");

#ifdef H3_PROF
// diagnostic build (profiles/variants.sh build kernels_conv_f16x3.hip h3prof "-DH3_PROF"): shader-clock sums per phase, wave 0 of every workgroup
__device__ unsigned long long g_h3prof[1024 * 8];
#define H3_STAMP(k) { const unsigned long long t_ = __builtin_readcyclecounter(); if (tid == 0) psum[k] += t_ - tlast; tlast = t_; }
#else
#define H3_STAMP(k)
#endif

// Input staging with ONE register per 16-byte chunk (the float32 kernel's Staging<> keeps three; this kernel is at the 256-register
// line of two workgroups per compute unit): chunk u of thread tid is tile pixel p = stage_pixel(tid) + 16 u = (row r, column c), channels
// 4 (tid & 15) ..; pk[u] = (r W + c) * pix | c -- pix is a multiple of 256, the low byte is free for the column, the only
// coordinate that needs a test (rows fall out of the buffer range by themselves) -- or -1: no such chunk.  (p: stage_pixel below.)
template <int DIL> struct StagingP { int pk[Geo<DIL>::XU]; };
// which of the 16 pixels of a chunk row thread tid stages: the two 16-lane groups of a 32-lane half take pixels 8 apart -- 8 x 272
// bytes = 32 banks: their 8-byte LDS writes (hi and lo halves of four channels) do not collide
__device__ __forceinline__ int stage_pixel(int tid) { const int g = tid >> 4; return (g >> 1) + 8 * (g & 1); }
template <int DIL>
__device__ __forceinline__ void staging_init_p(const ConvArgs& a, int tid, StagingP<DIL>& st, const int pix) {
    constexpr int HX = Geo<DIL>::HX, HY = Geo<DIL>::HY;
#pragma unroll
    for (int u = 0; u < Geo<DIL>::XU; ++u) {
        const int p = stage_pixel(tid) + 16 * u, r = p / HX, c = p - r * HX;
        st.pk[u] = (p < HY * HX) ? (((r * a.W + c) * pix) | c) : -1;
    }
}
// The request for an input tile is cut into pieces that are issued one per tap: vector-memory results return in order, so a wait for
// the weights of the next tap also waits for every older request -- a whole tile requested at once in front of a tap would be
// waited for one tap later.  FetchP holds what the pieces share.
struct FetchP { __amdgpu_buffer_rsrc_t rs; int origin, xlo, xhi; };
template <int DIL>
__device__ __forceinline__ FetchP fetch_begin(const ConvArgs& a, const TilePos& q, int tid, const int pix, const int coff, const bool any = true) {
    FetchP f;
    // any = false: a descriptor of zero bytes -- every piece is still ISSUED (hipcc counts the loads in flight exactly only when
    // they are unconditional: behind an `if` it waits for all of them at the next wait) but none reaches memory
    f.rs = image_rsrc(a.x + (size_t)q.img * a.H * a.W * (pix >> 2), any ? a.H : 0, a.W, pix, any ? coff : 0);
    f.origin = ((q.y0 - DIL) * a.W + (q.x0 - DIL)) * pix + 16 * (tid & 15);    // may be negative: such offsets are out of range as unsigned
    f.xlo = DIL - q.x0; f.xhi = a.W + DIL - q.x0;                              // valid tile columns: xlo <= c < xhi
    return f;
}
template <int DIL, int U0, int U1>
__device__ __forceinline__ void fetch_piece(const FetchP& f, const StagingP<DIL>& st, f32x4 (&v)[Geo<DIL>::XU]) {
#pragma unroll
    for (int u = U0; u < U1 && u < Geo<DIL>::XU; ++u) {
        const int c = st.pk[u] & 255;
        const bool in = st.pk[u] >= 0 && c >= f.xlo && c < f.xhi;
        const int off = in ? f.origin + (st.pk[u] & ~255) : -16;
        const u32x4v w = __builtin_amdgcn_raw_buffer_load_b128(f.rs, off, 0, 0);
        v[u] = f32x4{__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w)};
    }
}
template <int DIL>
__device__ __forceinline__ void fetch_input_p(const ConvArgs& a, const TilePos& q, const StagingP<DIL>& st, f32x4 (&v)[Geo<DIL>::XU],
                                              int tid, const int pix, const int coff) {
    const FetchP f = fetch_begin<DIL>(a, q, tid, pix, coff);
    fetch_piece<DIL, 0, Geo<DIL>::XU>(f, st, v);
}

// registers -> LDS tile, split on the way: chunk u of thread tid = 4 consecutive channels cq of tile pixel stage_pixel(tid) + 16 u --
// one base address per thread, the chunks are immediate offsets
// in_split (uniform): the tensor is in the SPLIT activation format (f16x3_common.h) -- the 16 bytes of chunk u are eight halves that go
// into the tile as they are: one ds_write_b128, no arithmetic
template <int DIL>
__device__ __forceinline__ void put_input_h3(float* xin, int tid, const f32x4 (&v)[Geo<DIL>::XU], const bool in_split = false) {
    if (in_split) {
        // chunk q = tid & 15 of the tensor's pixel: hi halves of channels 8 q .. (q < 8) or lo halves of channels 8 (q - 8) ..
        const int q = tid & 15;
        char* px = reinterpret_cast<char*>(xin) + stage_pixel(tid) * (CV_PS * 4) + h3_chunk_pos(q & 3, (q >> 2) & 1, q >> 3);
#pragma unroll
        for (int u = 0; u < Geo<DIL>::XU; ++u)
            if (stage_pixel(tid) + 16 * u < Geo<DIL>::HY * Geo<DIL>::HX)
                *reinterpret_cast<f32x4*>(px + u * (16 * CV_PS * 4)) = v[u];
        return;
    }
    // float32 input: this thread's four channels 4 t .. (t = tid & 15) are half of chunk (kb = (t >> 1) & 3, s2 = t >> 3)
    const int t = tid & 15;
    char* px = reinterpret_cast<char*>(xin) + stage_pixel(tid) * (CV_PS * 4) + h3_chunk_pos((t >> 1) & 3, t >> 3, 0) + 8 * (t & 1);
#pragma unroll
    for (int u = 0; u < Geo<DIL>::XU; ++u)
        if (stage_pixel(tid) + 16 * u < Geo<DIL>::HY * Geo<DIL>::HX) {
            h4 hi, lo;
            split4(v[u], hi, lo);
            *reinterpret_cast<h4*>(px + u * (16 * CV_PS * 4)) = hi;
            *reinterpret_cast<h4*>(px + u * (16 * CV_PS * 4) + 16) = lo;      // part 1 = the chunk behind part 0
        }
}

// The epilogue's second half when the skip input and / or the output are in the SPLIT format (store_rows32 of conv_common.h is the
// all-float32 one).  The wave's 32 pixels x 64 channels lie in `stage` in pixel order; a lane takes EIGHT consecutive channels
// (octet co = lane & 7) of pixel slot lane >> 3, four times: staged row m = 8 it + (lane >> 3) = tile row 2 w + (m >> 4), column m & 15.
// Eight channels are 32 bytes of a float32 pixel (two 16-byte accesses) or 16 bytes of hi halves + 16 bytes of lo halves, 128 bytes
// apart, of a split one: as many loads and stores per lane as the float32 epilogue issues.  y = relu?(staged + skip); the bias is
// already in the staged values (added when the accumulators were staged: four registers there instead of eight here).
template <int STR>
__device__ __forceinline__ void store_rows32_fmt(const ConvArgs& a, const TilePos& q, const float* stage, int wv, int lane,
                                                 const int pix, const int coff) {
    const __amdgpu_buffer_rsrc_t ry = image_rsrc(a.y + (size_t)q.img * a.H * a.W * (pix >> 2), a.H, a.W, pix, coff);
    const __amdgpu_buffer_rsrc_t rk = image_rsrc((a.skip ? a.skip : a.y) + (size_t)q.img * a.H * a.W * (pix >> 2), a.H, a.W, pix, coff);
    // (opaque copy of the lane number: everything derived from it is computed HERE, per item -- hoisted out of the item loop as the
    // loop invariants they are, these few values would live through the tap loop, which has no register to spare)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int ps = ln >> 3, co = ln & 7;
    const bool ksplit = (a.fmt & CV_FMT_SKIP) != 0, ysplit = (a.fmt & CV_FMT_Y) != 0;
    int pb[4];                                                   // byte offset of the pixel's 64-channel block, or out of range
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int m = 8 * it + ps, gx = q.x0 + (m & 15);
        pb[it] = gx < a.W ? ((q.y0 + 2 * wv + (m >> 4)) * a.W + gx) * pix : -256;      // rows below the image: beyond the buffer's range
    }
    // all eight skip requests first: one memory round trip
#pragma unroll
    for (int hh = 0; hh < 1; ++hh) {
        u32x4v k0[4], k1[4];
        if (a.skip) {
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                const int it = 4 * hh + e2;
                k0[e2] = __builtin_amdgcn_raw_buffer_load_b128(rk, pb[it] + (ksplit ? 16 * co : 32 * co), 0, 0);
                k1[e2] = __builtin_amdgcn_raw_buffer_load_b128(rk, pb[it] + (ksplit ? 16 * co + 128 : 32 * co + 16), 0, 0);
            }
        }
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
            const int it = 4 * hh + e2;
            const float* sp = stage + (8 * it + ps) * STR + 8 * co;
            f32x4 v0 = *reinterpret_cast<const f32x4*>(sp), v1 = *reinterpret_cast<const f32x4*>(sp + 4);
            if (a.skip) {
                if (ksplit) {
                    const h8 kh = __builtin_bit_cast(h8, k0[e2]), kl = __builtin_bit_cast(h8, k1[e2]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v0[e] += unsplit(kh[e], kl[e]); v1[e] += unsplit(kh[4 + e], kl[4 + e]); }
                } else {
                    v0 += f32x4{__uint_as_float(k0[e2].x), __uint_as_float(k0[e2].y), __uint_as_float(k0[e2].z), __uint_as_float(k0[e2].w)};
                    v1 += f32x4{__uint_as_float(k1[e2].x), __uint_as_float(k1[e2].y), __uint_as_float(k1[e2].z), __uint_as_float(k1[e2].w)};
                }
            }
            if (a.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v0[e] = relu_keep_nan(v0[e]); v1[e] = relu_keep_nan(v1[e]); }
            }
            if (ysplit) {
                h4 h0, l0, h1, l1;
                split4(v0, h0, l0);
                split4(v1, h1, l1);
                const h8 hi = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]}, lo = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, hi), ry, pb[it] + 16 * co, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, lo), ry, pb[it] + 16 * co + 128, 0, 0);
            } else {
                const u32x4v o0 = {__float_as_uint(v0[0]), __float_as_uint(v0[1]), __float_as_uint(v0[2]), __float_as_uint(v0[3])};
                const u32x4v o1 = {__float_as_uint(v1[0]), __float_as_uint(v1[1]), __float_as_uint(v1[2]), __float_as_uint(v1[3])};
                __builtin_amdgcn_raw_buffer_store_b128(o0, ry, pb[it] + 32 * co, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(o1, ry, pb[it] + 32 * co + 16, 0, 0);
            }
        }
    }
}

template <int DIL>
__global__ __launch_bounds__(CV_THREADS, GeoH<DIL>::WPS) void k_conv3x3_c64_h3(ConvArgs a, int nitems) {
    constexpr int HX = Geo<DIL>::HX;
    // ONE LDS array (input tile, then the two weight buffers): with LDS-DMA in flight hipcc orders accesses to separate arrays
    // conservatively
    __shared__ __attribute__((aligned(16))) float lds[Geo<DIL>::XIN + 2 * H3_TAP16 * 4];
    float* const xin = lds;
    f32x4 (*const wbuf)[H3_TAP16] = reinterpret_cast<f32x4 (*)[H3_TAP16]>(lds + Geo<DIL>::XIN);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // v_mfma_f32_16x16x32_f16: lane (i, kb) supplies A[row i][k = 8 kb ..] and B[k = 8 kb ..][column i]; the wave's 32 pixels are two
    // M tiles = its two tile rows (i = the pixel's column), its 64 output channels four N tiles
    const int i = lane & 15, kb = lane >> 4;
    // C = 64 NC channels: an ITEM is (tile, block cb of 64 output channels), its K loop runs over NC chunks of 64 input channels
    // x 9 taps.  item = tile * NC + cb and gridDim.x is a multiple of NC (launch): a workgroup keeps its cb, so its weight
    // stream -- [cb][chunk][tap] blocks of 16 KiB -- is periodic in 9 NC taps, and on an 8-XCD part every XCD works on NC / 8 ..
    // 1 blocks of output channels whose weights (147 KiB x NC) stay in ITS L2 while the input tiles stream.
    const int NC = a.C >> 6, pix = a.C * 4, period = 9 * NC;
    int item = blockIdx.x;
    if (item >= nitems) return;
    const int cb = item % NC;

    StagingP<DIL> st;
    staging_init_p<DIL>(a, tid, st, pix);
    f32x4 xpre[Geo<DIL>::XU];
    fetch_input_p<DIL>(a, tile_pos(a, item / NC), st, xpre, tid, pix, 0);
    // A tap's weights travel global memory -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write): wave w copies
    // units w * 64 + lane + 256 j of the 1024, one tap ahead of their use.  Completion is this wave's vmcnt; visibility to the
    // other waves the barrier after the wait.
    // (`buffer_load_dwordx4 ... offen lds` with the tap and piece in the SCALAR offset: no vector instruction forms an address)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, 9 * a.C * a.C * 4, 0x00020000);
    const int wvoff = tid * 16, wbase = cb * period * (H3_TAP16 * 16);
#define H3_DMA(buf_, t_)                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                           \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(&wbuf[buf_][wv * 64 + 256 * j]), 16, wvoff, \
                                                 wbase + (t_) * (H3_TAP16 * 16) + j * 4096, 0, 0);
    const bool in_split = (a.fmt & CV_FMT_X) != 0;               // uniform, the whole launch
    H3_DMA(0, 0)
    put_input_h3<DIL>(xin, tid, xpre, in_split);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int par = 0;                                                 // buffer of the current tap
    int t1 = 1 >= period ? 0 : 1;                                // stream position of the next tap's weights
    // the workgroup that arrived second on its SIMDs starts late, once (kernels_conv.hip): the two stay out of phase
    if (GeoH<DIL>::WPS > 1 && (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1)) __builtin_amdgcn_s_sleep(127);

#ifdef H3_PROF
    unsigned long long psum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
#pragma unroll 1
    for (; item < nitems; item += gridDim.x) {
        const TilePos q = tile_pos(a, item / NC);
        f32x4 mainv[2][4], corrv[2][4];                            // [M tile = tile row of the wave][N tile of 16 channels]
        // One chunk = 64 input channels x 9 taps.  The first chunk of an item is its own instance of the code (FIRST): there the first
        // MFMA of every accumulator takes the constant 0 as its C operand -- no 64 register moves per item to clear them.
        auto chunk = [&](const int cc, auto first_tag) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            // the input tile that follows this one -- the tile's next 64 input channels, or the first 64 of the next item -- is
            // requested now and consumed after this chunk's nine taps
            const bool last = cc + 1 == NC;
            const bool more = !last || item + (int)gridDim.x < nitems;
            H3_STAMP(0)
#ifndef H3_ABL_NOXPRE
            const FetchP nx = fetch_begin<DIL>(a, last && more ? tile_pos(a, (item + gridDim.x) / NC) : q, tid, pix, last ? 0 : 64 * (cc + 1), more);
#else
            const FetchP nx = fetch_begin<DIL>(a, q, tid, pix, 0, false);
#endif
            H3_STAMP(1)
            constexpr int PIECE = (Geo<DIL>::XU + 5) / 6;           // the next input tile: requested in six pieces, behind the weights of taps 0..5
            // One tap = 2 K steps of 32 input channels = 4 half steps (a K step x two of the four N tiles) of 12 MFMAs.  What a
            // wave's critical path sees of a tap besides its MFMAs is kept short: the A operands of K step 0 are read BEFORE the
            // barrier (the input tile does not change inside a chunk), only the B reads of half step 0 stand between the barrier
            // and the first MFMA, and the request for the next tap's weights leaves behind half step 0's MFMAs.
            h8 ah[2][2], al[2][2], bh[2][2], bl[2][2];                 // [slot][M tile] / [slot][N tile of the pair]
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#define H3_LOAD_A(slot, ap_, s2_)                                                                        \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) {                                           \
                ah[slot][mt] = *reinterpret_cast<const h8*>((ap_) + mt * (HX * CV_PS * 4) + 32 * (s2_));        \
                al[slot][mt] = *reinterpret_cast<const h8*>((ap_) + mt * (HX * CV_PS * 4) + 32 * (s2_) + 16);   \
            }
#define H3_LOAD_B(slot, h_)                                                                              \
            _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) {                                           \
                bh[slot][q_] = *reinterpret_cast<const h8*>(bp + 1024 * ((((h_) >> 1) * 4 + 2 * ((h_) & 1) + q_) * 2));      \
                bl[slot][q_] = *reinterpret_cast<const h8*>(bp + 1024 * ((((h_) >> 1) * 4 + 2 * ((h_) & 1) + q_) * 2 + 1));  \
            }
#define H3_MFMA(h_)                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                           \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                             \
            _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) {                                           \
                const int nt_ = 2 * ((h_) & 1) + q_;                                                     \
                const bool z_ = FIRST && tap == 0 && (h_) < 2;         /* compile-time: the accumulators' first use */   \
                mainv[mt][nt_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[((h_) >> 1) & 1][mt], bh[(h_) & 1][q_], z_ ? zero4 : mainv[mt][nt_], 0, 0, 0);  \
                corrv[mt][nt_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[((h_) >> 1) & 1][mt], bl[(h_) & 1][q_], z_ ? zero4 : corrv[mt][nt_], 0, 0, 0);  \
                corrv[mt][nt_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[((h_) >> 1) & 1][mt], bh[(h_) & 1][q_], corrv[mt][nt_], 0, 0, 0);  \
            }                                                                                            \
            __builtin_amdgcn_sched_barrier(0);
            // tap (0, 0), M tile 0: pixel (row 2 w, column h3_row_pixel(i)); + HX pixels: M tile 1; + 32 s2: K step; + 16: the lo halves
            // (the conflict-free chunk order of f16x3_common.h)
            const char* const a0 = reinterpret_cast<const char*>(xin) + (2 * wv * HX + h3_row_pixel(i)) * (CV_PS * 4) + h3_chunk_pos(kb, 0, 0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - 3 * ky;
                const char* ap = a0 + (ky * DIL * HX + kx * DIL) * (CV_PS * 4);                     // input pixel of this tap
                const char* bp = reinterpret_cast<const char*>(&wbuf[par][0]) + lane * 16;          // + 1024 f: fragment f = ((s2 * 4 + nt) * 2 + part
                // this wave's DMA of this tap's weights has landed (the pieces of the input prefetch issued behind it may still be
                // in flight: a counted wait), its LDS reads are back; then the barrier -- raw: __syncthreads() would drain vmcnt
#define H3_WAIT_BUT(n_) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n_) : "memory");
#define H3_PIECE_LOADS(k_) ((k_) * PIECE >= Geo<DIL>::XU ? 0 : ((k_) + 1) * PIECE <= Geo<DIL>::XU ? PIECE : Geo<DIL>::XU - (k_) * PIECE)
                if (tap == 0 || tap >= 7) { H3_WAIT_BUT(0) }          // nothing was requested behind the tap before's DMA
                if (tap == 1) { H3_WAIT_BUT(H3_PIECE_LOADS(0)) }
                if (tap == 2) { H3_WAIT_BUT(H3_PIECE_LOADS(1)) }
                if (tap == 3) { H3_WAIT_BUT(H3_PIECE_LOADS(2)) }
                if (tap == 4) { H3_WAIT_BUT(H3_PIECE_LOADS(3)) }
                if (tap == 5) { H3_WAIT_BUT(H3_PIECE_LOADS(4)) }
                if (tap == 6) { H3_WAIT_BUT(H3_PIECE_LOADS(5)) }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef H3_ABL_NOBAR                                                  // H3_ABL_*: timing-only ablations (results wrong by design), profiles/experiments/abl_conv_f16x3.sh
                __builtin_amdgcn_s_barrier();
#endif
                asm volatile("" ::: "memory");
                H3_STAMP(2)
                if (tap == 0) { H3_LOAD_A(0, ap, 0) }                 // taps 1..8: read at the end of the tap before
                H3_LOAD_B(0, 0)
                H3_LOAD_B(1, 1)
                H3_MFMA(0)
#ifndef H3_ABL_NOWWRITE
                H3_DMA(par ^ 1, t1)                                  // the next tap's weights
#endif
                // the counted wait of the next tap (vmcnt = the piece's loads) is right only if the four DMAs are OLDER than the piece:
                // pin the order (the two groups do not alias, hipcc is otherwise free to interleave them); tools/isa_scan.py checks it
                __builtin_amdgcn_sched_barrier(0);
#ifndef H3_ABL_WSAME
                t1 = t1 + 1 == period ? 0 : t1 + 1;
#endif
                if (tap == 0) fetch_piece<DIL, 0 * PIECE, 1 * PIECE>(nx, st, xpre);
                if (tap == 1) fetch_piece<DIL, 1 * PIECE, 2 * PIECE>(nx, st, xpre);
                if (tap == 2) fetch_piece<DIL, 2 * PIECE, 3 * PIECE>(nx, st, xpre);
                if (tap == 3) fetch_piece<DIL, 3 * PIECE, 4 * PIECE>(nx, st, xpre);
                if (tap == 4) fetch_piece<DIL, 4 * PIECE, 5 * PIECE>(nx, st, xpre);
                if (tap == 5) fetch_piece<DIL, 5 * PIECE, 6 * PIECE>(nx, st, xpre);
                H3_STAMP(3)
                H3_LOAD_A(1, ap, 1)
                H3_LOAD_B(0, 2)
                H3_MFMA(1)
                H3_LOAD_B(1, 3)
                H3_MFMA(2)
                if (tap + 1 < 9) {
                    const int ky1 = (tap + 1) / 3, kx1 = tap + 1 - 3 * ky1;
                    H3_LOAD_A(0, a0 + (ky1 * DIL * HX + kx1 * DIL) * (CV_PS * 4), 0)
                }
                H3_MFMA(3)
                par ^= 1;
                H3_STAMP(4)
            }
#undef H3_LOAD_A
#undef H3_LOAD_B
#undef H3_MFMA
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every wave is done with this chunk's input (raw barrier: the DMA of
            __builtin_amdgcn_s_barrier();                            // the next tap's weights stays in flight)
            asm volatile("" ::: "memory");
            H3_STAMP(5)
            if (last) {
                // accumulator (reg r, lane (i, kb)) of tile (mt, nt) = M-tile row 4 kb + r = pixel (tile row mt of the wave, column h3_row_pixel(4 kb + r)), channel 16 nt + i:
                // into the wave's staging rows (pixel order, 68 floats apart: the four lane groups of a write start 16 banks apart)
                float* stage = xin + wv * (32 * H3_STR);
#ifndef H3_ABL_NOEPI
                const bool fmt_epi = (a.fmt & (CV_FMT_SKIP | CV_FMT_Y)) != 0;
                float bs[4] = {0.f, 0.f, 0.f, 0.f};                    // the split-format epilogue takes the bias here (this lane's four channels)
                if (fmt_epi && a.bias) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) bs[nt] = a.bias[64 * cb + 16 * nt + i];
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            stage[(16 * mt + h3_row_pixel(4 * kb) + r) * H3_STR + 16 * nt + i] = fmaf(corrv[mt][nt][r], H3_RSCALE, mainv[mt][nt][r]) + bs[nt];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // a wave's LDS instructions execute in order: compiler-only ordering
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (fmt_epi) store_rows32_fmt<H3_STR>(a, q, stage, wv, lane, pix, 64 * cb);
                else store_rows32<H3_STR>(a, q, stage, wv, lane, 0, pix, 64 * cb, a.bias ? a.bias + 64 * cb : nullptr);
#else
                if (mainv[0][0][0] + corrv[1][3][3] == 123.456f) a.y[tid] = 1.f;
#endif
                H3_STAMP(6)
                if (more) __syncthreads();                           // every wave is done with the staging area
            }
#ifndef H3_ABL_NOPUT
            if (more) put_input_h3<DIL>(xin, tid, xpre, in_split);     // published by the barrier of the next chunk's first tap
#else
            if (more && xpre[0][0] + xpre[5][1] + xpre[11][2] == 123.456f) a.y[tid] = 2.f;
#endif
            H3_STAMP(7)
        };
        chunk(0, std::true_type{});
#pragma unroll 1
        for (int cc = 1; cc < NC; ++cc) chunk(cc, std::false_type{});
    }
    // the last tap requested one more block of weights: no wave ends with an LDS-DMA in flight (its LDS may belong to the next
    // workgroup by the time the data lands)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef H3_PROF
    if (tid == 0 && blockIdx.x < 1024) for (int k = 0; k < 8; ++k) g_h3prof[blockIdx.x * 8 + k] = psum[k];
#endif
}

#ifdef H3_PROF
extern "C" int pnp_conv_h3_prof_read(unsigned long long* out /* [1024][8] */) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_h3prof), sizeof(unsigned long long) * 1024 * 8);
}
#endif

// torch.nn.Conv2d weight [C out][C in][3][3] -> split halves in fragment order, blocks [cb][chunk cc][tap] of 16 KiB: half j of lane
// (n, kb) of fragment (K step s, N tile nt, part) is part(W[out = 64 cb + 16 nt + n][in = 64 cc + 32 s + 8 kb + j][tap]) --
// v_mfma_f32_16x16x32_f16: lane l supplies B[k = 8 (l >> 4) + j][column l & 15]; the A side reads input channels in the same
// order.  As many bytes as the float32 weights.  Once per model.
__global__ __launch_bounds__(256) void k_conv_pack_w_h3(const float* w_oihw, _Float16* wfrag, int C) {
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;      // one (hi, lo) pair per thread
    if (o >= 9LL * C * C) return;
    const int NC = C >> 6;
    const int j = o & 7, lane = (o >> 3) & 63, nt = (o >> 9) & 3, s = (o >> 11) & 1;
    const long long blk = o >> 12;                                // (cb * NC + cc) * 9 + tap
    const int tap = (int)(blk % 9), cc = (int)((blk / 9) % NC), cb = (int)(blk / (9 * NC));
    const int out = 64 * cb + 16 * nt + (lane & 15), in = 64 * cc + 32 * s + 8 * (lane >> 4) + j;
    const float w = w_oihw[((size_t)out * C + in) * 9 + tap];
    const _Float16 hi = (_Float16)w;
    const _Float16 lo = (_Float16)((w - (float)hi) * H3_SCALE);
    const size_t frag = ((size_t)blk * 2 + s) * 8 + nt * 2;      // the hi fragment; lo follows
    wfrag[(frag * 64 + lane) * 8 + j] = hi;
    wfrag[((frag + 1) * 64 + lane) * 8 + j] = lo;
}

// ------------------------------------------------------------------------------------------
// Last layer of the stacks (64 -> COUT <= 4 channels, NHWC in, NCHW out, + bias; models/network_ffdnet.py:56, network_dncnn.py:62) in
// the same arithmetic.  On the vector units (kernels_conv.hip: k_conv3x3_tail) this layer costs as much as a body layer of the
// f16x3 kernel; here it is a 16-column matrix product of which COUT columns are used: per wave two M tiles (its two tile rows) x
// one N tile, 18 K steps of 32 x 3 products = 108 MFMAs per tile -- the layer is bound by reading its input.  One workgroup per
// 8 x 16 tile, 58 KiB of LDS (the split input tile + the split weights of COUT channels): two workgroups per compute unit.
// ------------------------------------------------------------------------------------------
struct TailH3Args {
    const float* x; const float* x2; const float* w; const float* bias; float* y;      // x2: null, or a tensor of x's shape added to it (the U-Net's last skip sum)
    int n, cout, H, W, tiles_x, tiles_y;
    // FFDNet's output stage folded into the stores (models/network_ffdnet.py:70-73): cout = 4, and channel 2 dy + dx of pixel (y, x) is pixel
    // (2 y + dy, 2 x + dx) of the ONE-channel full-resolution result y [n][1][out_h][out_w] (pixel shuffle + the crop of the padded row / column)
    int shuffle, out_h, out_w;
};
__global__ __launch_bounds__(CV_THREADS, 2) void k_conv3x3_tail_h3(TailH3Args t, int ntiles) {
    // PERSISTENT since round 6 (two workgroups per compute unit, tile = blockIdx.x + k gridDim.x): the next tile's input is requested into
    // registers before this tile's matrix products and split into LDS behind them -- round 5's kernel was one workgroup per tile with nothing
    // to overlap its fetch (130 us for 268 MB at 64 slices of 128 x 128: twice the memory time).
    constexpr int HX = Geo<1>::HX;
    __shared__ __attribute__((aligned(16))) float xin[Geo<1>::XIN];
    __shared__ __attribute__((aligned(16))) _Float16 wl[9 * 2 * 2 * 4 * 4 * 8];      // [tap][K step][hi, lo][kb][n < 4][8 halves]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kb = lane >> 4;
    int tile = blockIdx.x;
    if (tile >= ntiles) return;                                   // (uniform)
    ConvArgs a;                                                   // the staging helpers' view of the input
    a.x = t.x; a.H = t.H; a.W = t.W; a.tiles_x = t.tiles_x; a.tiles_y = t.tiles_y; a.fmt = 0;
    StagingP<1> st;
    staging_init_p<1>(a, tid, st, CV_C * 4);
    f32x4 xpre[Geo<1>::XU];
    auto fetch = [&](const TilePos& q) __attribute__((always_inline)) {
        a.x = t.x;
        fetch_input_p<1>(a, q, st, xpre, tid, CV_C * 4, 0);
        if (t.x2) {                                               // uniform: x + x2, the sum never goes to memory (models/network_unet.py:134)
            a.x = t.x2;
            f32x4 x2pre[Geo<1>::XU];
            fetch_input_p<1>(a, q, st, x2pre, tid, CV_C * 4, 0);
#pragma unroll
            for (int u = 0; u < Geo<1>::XU; ++u) xpre[u] += x2pre[u];
        }
    };
    fetch(tile_pos(a, tile));
    // weights: w_oihw [cout][64][3][3] -> split halves in the B operand's order (columns >= cout are zeros the lanes supply themselves); once
    for (int e = tid; e < 9 * 2 * 4 * 4 * 8; e += CV_THREADS) {
        const int j = e & 7, n = (e >> 3) & 3, kq = (e >> 5) & 3, s2 = (e >> 7) & 1, tap = e >> 8;
        const float w = n < t.cout ? t.w[((size_t)n * 64 + 32 * s2 + 8 * kq + j) * 9 + tap] : 0.f;
        const _Float16 hi = (_Float16)w;
        const int base = (((tap * 2 + s2) * 2) * 4 + kq) * 32 + n * 8 + j;
        wl[base] = hi;
        wl[base + 4 * 32] = (_Float16)((w - (float)hi) * H3_SCALE);
    }
    const char* const a0 = reinterpret_cast<const char*>(xin) + (2 * wv * HX + h3_row_pixel(i)) * (CV_PS * 4) + h3_chunk_pos(kb, 0, 0);
    const char* const b0 = reinterpret_cast<const char*>(wl) + kb * 64 + (i & 3) * 16;
    const bool col = i < t.cout;
    const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    const float b = (col && t.bias) ? t.bias[i] : 0.f;
#pragma unroll 1
    for (; tile < ntiles; tile += gridDim.x) {
        const TilePos q = tile_pos(a, tile);
        put_input_h3<1>(xin, tid, xpre);
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile_pos(a, tile + gridDim.x));       // in flight behind this tile's matrix products
        f32x4 mainv[2], corrv[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) { mainv[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; corrv[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const char* ap = a0 + (ky * HX + kx) * (CV_PS * 4);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                h8 bh = *reinterpret_cast<const h8*>(b0 + ((tap * 2 + s2) * 2) * 256);
                h8 bl = *reinterpret_cast<const h8*>(b0 + ((tap * 2 + s2) * 2 + 1) * 256);
                bh = col ? bh : zero; bl = col ? bl : zero;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const h8 ah = *reinterpret_cast<const h8*>(ap + mt * (HX * CV_PS * 4) + 32 * s2);
                    const h8 al = *reinterpret_cast<const h8*>(ap + mt * (HX * CV_PS * 4) + 32 * s2 + 16);
                    mainv[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, mainv[mt], 0, 0, 0);
                    corrv[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, corrv[mt], 0, 0, 0);
                    corrv[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, corrv[mt], 0, 0, 0);
                }
            }
        }
        // accumulator (reg r, lane (i, kb)) of M tile mt = pixel (tile row 2 w + mt, column h3_row_pixel(4 kb + r)), output channel i
        __syncthreads();                                          // every wave is past its taps: the tile may be reused (below, or by the next tile)
        if (t.shuffle) {
            // FFDNet: the tile's 8 x 16 x 4 values are a 16 x 32 block of the full-resolution result; it is assembled in LDS and leaves as
            // whole rows, two consecutive pixels per thread
            if (col) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        xin[(2 * (2 * wv + mt) + (i >> 1)) * 32 + 2 * (h3_row_pixel(4 * kb) + r) + (i & 1)] = fmaf(corrv[mt][r], H3_RSCALE, mainv[mt][r]) + b;
            }
            __syncthreads();
            const int orow = tid >> 4, ocol = 2 * (tid & 15), oy = 2 * q.y0 + orow, ox = 2 * q.x0 + ocol;
            if (oy < t.out_h) {
                float* dst = t.y + ((size_t)q.img * t.out_h + oy) * t.out_w + ox;
                if (ox < t.out_w) dst[0] = xin[orow * 32 + ocol];
                if (ox + 1 < t.out_w) dst[1] = xin[orow * 32 + ocol + 1];
            }
            __syncthreads();                                      // the assembled block is read: the next tile may be written
        } else if (col) {
            const size_t plane = (size_t)t.H * t.W;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int gy = q.y0 + 2 * wv + mt;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gx = q.x0 + h3_row_pixel(4 * kb) + r;
                    if (gy < t.H && gx < t.W)
                        t.y[((size_t)q.img * t.cout + i) * plane + (size_t)gy * t.W + gx] = fmaf(corrv[mt][r], H3_RSCALE, mainv[mt][r]) + b;
                }
            }
        }
    }
}

hipError_t launch_conv3x3_tail_f16x3(hipStream_t s, const float* x_nhwc, const float* x2_nhwc, const float* w_oihw, const float* bias, float* y_nchw,
                                     int n, int cout, int H, int W, int shuffle_h, int shuffle_w) {
    if (cout < 1 || cout > 4 || (long long)H * W * CV_C * 4 > 0x7fffffffLL) return hipErrorInvalidValue;
    if (shuffle_h && (cout != 4 || (shuffle_h + 1) / 2 != H || (shuffle_w + 1) / 2 != W)) return hipErrorInvalidValue;
    TailH3Args t;
    t.shuffle = shuffle_h ? 1 : 0; t.out_h = shuffle_h; t.out_w = shuffle_w;
    t.x = x_nhwc; t.x2 = x2_nhwc; t.w = w_oihw; t.bias = bias; t.y = y_nchw; t.n = n; t.cout = cout; t.H = H; t.W = W;
    t.tiles_x = (W + CV_TX - 1) / CV_TX; t.tiles_y = (H + CV_TY - 1) / CV_TY;
    const long long tiles = (long long)n * t.tiles_x * t.tiles_y;
    if (tiles <= 0 || tiles > 0x7fffffffLL) return hipErrorInvalidValue;
    const int cus = conv_compute_units();
    if (cus <= 0) return hipGetLastError();
    const long long grid = tiles < 2LL * cus ? tiles : 2LL * cus;          // persistent, two workgroups per compute unit; every loop ends: tile < ntiles
    hipLaunchKernelGGL(k_conv3x3_tail_h3, dim3((unsigned)grid), dim3(CV_THREADS), 0, s, t, (int)tiles);
    return hipGetLastError();
}

template <int DIL>
static hipError_t launch_h3_dil(hipStream_t s, const ConvArgs& a, long long items, int cus) {
    // persistent workgroups, a multiple of NC = C / 64 of them (a workgroup keeps its block of output channels); every
    // workgroup's loop ends: item < nitems
    const int NC = a.C >> 6;
    long long grid = (long long)GeoH<DIL>::WPS * cus;
    grid -= grid % NC;
    if (grid < NC) grid = NC;
    if (items < grid) grid = items;                               // items = tiles * NC: a multiple of NC as well
    hipLaunchKernelGGL(k_conv3x3_c64_h3<DIL>, dim3((unsigned)grid), dim3(CV_THREADS), 0, s, a, (int)items);
    return hipGetLastError();
}

hipError_t launch_conv3x3_f16x3(hipStream_t s, const float* x, const float* w, const float* bias, const float* skip, float* y,
                                int n, int C, int H, int W, int relu, int dilation, int fmt) {
    if (C < 64 || C > 1024 || (C & 63) || (C != 64 && dilation != 1) || (fmt & ~(CV_FMT_X | CV_FMT_SKIP | CV_FMT_Y))) return hipErrorInvalidValue;
    ConvArgs a;
    a.x = x; a.w = w; a.bias = bias; a.skip = skip; a.y = y; a.n = n; a.H = H; a.W = W; a.relu = relu; a.C = C; a.fmt = fmt;
    a.tiles_x = (W + CV_TX - 1) / CV_TX; a.tiles_y = (H + CV_TY - 1) / CV_TY;
    const long long items = (long long)n * a.tiles_x * a.tiles_y * (C >> 6);
    if (items <= 0 || items > 0x7fffffffLL) return hipErrorInvalidValue;
    if ((long long)H * W * C * 4 > 0x7fffffffLL) return hipErrorInvalidValue;           // one image must fit a signed 32-bit buffer offset
    const int cus = conv_compute_units();
    if (cus <= 0) return hipGetLastError();
    if (dilation == 1) {
        // the wide kernel (kernels_conv_f16x3_wide.hip: 16 x 16 tiles, ONE workgroup per compute unit) once every compute unit has an item of
        // its own -- below that the 8 x 16 tiles on two workgroups per unit spread a small layer better (one 256 x 256 slice at 64 channels:
        // 17.1 us wide against 18.2 narrow; one 128 x 128 slice: 13.4 against 10.3, profiles/conv_f16x3_wide_probe_r06.txt)
        const long long items16 = (long long)n * ((W + 15) / 16) * ((H + 15) / 16) * (C >> 6);
        const int mode = conv_wide_mode();
        if (mode >= 1 || (mode < 0 && items16 >= (long long)cus)) return launch_conv3x3_f16x3_wide(s, x, w, bias, skip, y, n, C, H, W, relu, fmt);
    }
    switch (dilation) {
        case 1: return launch_h3_dil<1>(s, a, items, cus);
        case 2: return launch_h3_dil<2>(s, a, items, cus);
        case 3: return launch_h3_dil<3>(s, a, items, cus);
        case 4: return launch_h3_dil<4>(s, a, items, cus);
        default: return hipErrorInvalidValue;
    }
}

static int& wide_mode_ref() {
    static int mode = [] { const char* e = getenv("PNP_CONV_WIDE"); return e ? atoi(e) : -1; }();
    return mode;
}
int conv_wide_mode() { return wide_mode_ref(); }
int conv_set_wide_mode(int m) { int& r = wide_mode_ref(); const int old = r; r = m < 0 ? -1 : m > 0 ? 1 : 0; return old; }

hipError_t launch_conv_pack_w_f16x3(hipStream_t s, const float* w_oihw, float* wfrag, int C) {
    if (C < 64 || C > 1024 || (C & 63)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_conv_pack_w_h3, dim3((unsigned)(9LL * C * C / 256)), dim3(256), 0, s, w_oihw, reinterpret_cast<_Float16*>(wfrag), C);
    return hipGetLastError();
}

}  // namespace pnp
