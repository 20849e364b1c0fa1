// The wide split-half conv3x3 kernel (kernels_conv_f16x3_wide.hip) for 64 -> 64 channels without a skip input -- the body layers of FFDNet, DnCNN,
// FDnCNN, IRCNN at dilation 1, the first convolution of DRUNet's full-resolution residual blocks -- with HALF OF THE EPILOGUE INSIDE THE TAP LOOP:
// "pipe".  Round 6; models/basicblock.py:63-100, network_ffdnet.py:58-73, network_dncnn.py:36-67.
//
// Why.  In the wide kernel an item of 64 input channels is 13.8 k cycles of MFMAs per compute wave and 4.6 k cycles of epilogue during which
// the matrix pipe idles (profiles/conv_f16x3_wide_r06.txt): 2.6 k of vector instructions and store issue, 1.7 k of waiting for the memory side to
// take a 64 KiB burst per compute unit (profiles/experiments/conv_f16x3_wide_epilogue_ablations_r06.txt).  A second accumulator set to overlap it
// with the next item does not fit (128 + 128 registers).  Here the 64 output channels of a wave are computed in TWO PASSES over the nine taps:
// pass A the channels of tile pair g = 0 (64 accumulator registers), pass B those of g = 1 (the other 64).  While pass B's MFMAs run, the
// results of pass A are finished and stored piece by piece BETWEEN them (one tile row of pixels every other tap: ~56 vector instructions and two
// stores inside 48 MFMAs, whose issue slots are half empty); what remains behind the last tap is pass B's half, and it overlaps the helper
// waves' hand-over of the next input tile, which is bound by LDS store bandwidth (88 KB, ~1.2 k cycles) and had been hidden by the epilogue.
// Price: the pixel fragments are read for both passes (one LDS read per two MFMAs, the narrow kernel's ratio -- the LDS has the room), a barrier
// per 48 MFMAs instead of per 96.  A (tap, pass) block of weights is 8 KiB: four buffers = 32 KiB of LDS.
// Everything else is the wide kernel's: 16 x 16 pixel tiles, compute waves 0-3 / helper waves 4-7, the permuted weight-row read, stores straight
// from the accumulators, bit-equal results (tests/test_gpu_conv.py).
#include "f16x3_wide_common.h"

namespace pnp {

constexpr int WP_PT = 18;                                        // pass-taps per item: (pass g, tap) = (p / 9, p % 9)
constexpr int WP_WB = 8192;                                      // bytes of one (tap, pass) block of weights: [K step s2][tile e of the pair][hi, lo][lane]
constexpr int WP_NBUF = 4;
#ifndef WP_MIX_VALU
#define WP_MIX_VALU 2                                            // vector instructions of a piece per MFMA of a mixed pass-tap
#endif

// one piece of the epilogue: the lane's eight channels 32 g + 8 kb .. of pixel (tile row 4 w + pt, column h3_row_pixel(i)) -- y = relu?(acc + bias)
template <bool YSPLIT>
__device__ __forceinline__ void wp_piece(const f32x4 (&mainv)[2][4], const f32x4 (&corrv)[2][4], const int pt, const f32x4 (&bs)[2], const float thr,
                                         const __amdgpu_buffer_rsrc_t ry, const int off /* byte offset of the lane's first channel (float32) or hi half (split) */) {
    f32x4 v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[e][r] = fmaf(corrv[e][pt][r], H3_RSCALE, mainv[e][pt][r]);
        v[e] += bs[e];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[e][r] = v[e][r] < thr ? 0.f : v[e][r];
    }
    if (YSPLIT) {
        h4 h0, l0, h1, l1;
        split4(v[0], h0, l0);
        split4(v[1], h1, l1);
        const h8 hi = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]}, lo = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, hi), ry, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, lo), ry, off + 128, 0, 0);
    } else {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const u32x4v o = {__float_as_uint(v[e][0]), __float_as_uint(v[e][1]), __float_as_uint(v[e][2]), __float_as_uint(v[e][3])};
            __builtin_amdgcn_raw_buffer_store_b128(o, ry, off + 16 * e, 0, 0);
        }
    }
}

#define WP_PIECE_LOADS(p_) ((p_) < 10 ? 2 : (p_) == 10 ? 1 : 0)    // the next input tile: 21 chunks per helper thread, two per pass-tap

template <bool YSPLIT>
__global__ __launch_bounds__(WT_THREADS, 2) void k_conv3x3_h3p(ConvArgs a, int nitems) {
    // ONE LDS array: the input tile, the four weight buffers, the biases
    __shared__ __attribute__((aligned(16))) float lds[WT_XIN + WP_NBUF * (WP_WB / 4) + 64];
    float* const xin = lds;
    char* const wbuf = reinterpret_cast<char*>(lds + WT_XIN);
    float* const lbias = lds + WT_XIN + WP_NBUF * (WP_WB / 4);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pix = 256;                                         // 64 channels
    int item = blockIdx.x;
    if (item >= nitems) return;                                  // (uniform over the workgroup)
    const bool in_split = (a.fmt & CV_FMT_X) != 0;

    // BARRIER PLAN -- identical for the eight waves: per item eighteen barriers B_p, one in front of every pass-tap, and one barrier E behind the
    // last.  At B_P (global pass-tap counter P): the weight blocks of P and P + 1 have landed in buffers P % 4, (P + 1) % 4; every compute wave
    // is done with P - 1, so buffer (P + 3) % 4 may be overwritten; at B_0 of an item the input tile is in place.  At E every compute wave is
    // done with the input tile.  An item has 18 pass-taps: the buffer of pass-tap p is (rb + p) % 4 with rb toggling 0 / 2 per item.
    if (wv >= 4) {
        // ------------------------------------------------ HELPER waves (4..7) ------------------------------------------------
        const int h = tid - WT_HTHREADS, hw = wv - 4;
        WStaging st;
        wstaging_init(a, h, st, pix);
        f32x4 xpre[WT_XU];
        {
            const WFetch f0 = wfetch_begin(a, wtile_pos(a, item), h, pix, 0);
            wfetch_piece<0, WT_XU>(f0, st, xpre);
        }
        // A (tap, pass g) block of weights: the rows of the packed tap's fragments nt = 2 g, 2 g + 1, both K steps, copied by LDS-DMA in the order
        // the compute waves want them: LDS unit (s2, e, part) = 1 KiB = lane (i, kb)'s fragment of tile e of the pair, whose row i is output
        // channel 32 g + 8 (i >> 2) + 4 e + (i & 3) (f16x3_wide_common.h) -- the DMA's per-lane GLOBAL address does the permutation, so the
        // compute waves read at lane * 16 with no bank conflict.  Helper wave w copies units (s2, e = w >> 1, part = w & 1), s2 = 0, 1.
        // Pass-tap p requests the block of p + 3; the block's position goes in the SCALAR offset.
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, 9 * 64 * 64 * 4, 0x00020000);
        const int wvoff = wt_wrow_offset(lane & 15, lane >> 4);
        char* const wb0 = wbuf + hw * 1024;
#define WP_GOFF(q_) (((q_) % 9) * 16384 + ((q_) / 9) * 4096 + (hw & 1) * 1024 + (hw >> 1) * 64)
#define WP_DMA(rot_, q_)                                                                                                      \
        _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2)                                                                       \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(wb0 + (rot_) * WP_WB + s2 * 4096), 16, wvoff, \
                                                     WP_GOFF(q_) + s2 * 8192, 0, 0);
        WP_DMA(0, 0) WP_DMA(1, 1) WP_DMA(2, 2)
        wput_input(xin, h, xpre, in_split);
        {   // the biases, read by the compute waves' epilogues from LDS (no bias: a descriptor of zero bytes returns zeros)
            const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias ? a.bias : a.w), 0, a.bias ? 256 : 0, 0x00020000);
            const float bv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb_, (h & 63) * 4, 0, 0));
            if (h < 64) lbias[h] = bv;
        }
        WT_WAIT_VM(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // raw barriers: this wave's LDS writes are done before it arrives
        int rb = 0;
#pragma unroll 1
        for (; item < nitems; item += gridDim.x) {
            const bool more = item + (int)gridDim.x < nitems;
            const WFetch nx = wfetch_begin(a, wtile_pos(a, more ? item + gridDim.x : item), h, pix, 0, more);
#pragma unroll
            for (int p = 0; p < WP_PT; ++p) {
                __builtin_amdgcn_s_barrier();                     // B_p
                asm volatile("" ::: "memory");
                WP_DMA((rb + p + 3) & 3, (p + 3) % WP_PT)         // the block of pass-tap P + 3 into the buffer P - 1 has left
                // the counted wait below is right only if the two DMAs are OLDER than the piece: pin the order (tools/isa_scan.py checks it)
                __builtin_amdgcn_sched_barrier(0);
                if (p == 0) wfetch_piece<0, 2>(nx, st, xpre);
                if (p == 1) wfetch_piece<2, 4>(nx, st, xpre);
                if (p == 2) wfetch_piece<4, 6>(nx, st, xpre);
                if (p == 3) wfetch_piece<6, 8>(nx, st, xpre);
                if (p == 4) wfetch_piece<8, 10>(nx, st, xpre);
                if (p == 5) wfetch_piece<10, 12>(nx, st, xpre);
                if (p == 6) wfetch_piece<12, 14>(nx, st, xpre);
                if (p == 7) wfetch_piece<14, 16>(nx, st, xpre);
                if (p == 8) wfetch_piece<16, 18>(nx, st, xpre);
                if (p == 9) wfetch_piece<18, 20>(nx, st, xpre);
                if (p == 10) wfetch_piece<20, 21>(nx, st, xpre);
                __builtin_amdgcn_sched_barrier(0);
                // everything but this pass-tap's two DMAs and its piece has completed: the block of P + 2 (requested one pass-tap ago) is in place
                if (p < 10) { WT_WAIT_VM(2 + 2); }
                if (p == 10) { WT_WAIT_VM(2 + 1); }
                if (p > 10) { WT_WAIT_VM(2); }
            }
            __builtin_amdgcn_s_barrier();                         // E: the compute waves are done with the input tile
            asm volatile("" ::: "memory");
            if (more) wput_input(xin, h, xpre, in_split);         // published by B_0 of the next item
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            rb ^= 2;
        }
        // no wave ends with an LDS-DMA in flight (its LDS may belong to the next workgroup by the time the data lands)
        WT_WAIT_VM(0);
#undef WP_DMA
#undef WP_GOFF
        return;
    }

    // ------------------------------------------------ COMPUTE waves ------------------------------------------------
    // v_mfma_f32_16x16x32_f16, D = A B with A = the weights (row = output channel, permuted: f16x3_wide_common.h) and B = the pixels.
    // Accumulator reg r of lane (i, kb), pass g, tile (e, pt) = channel 32 g + 8 kb + 4 e + r of pixel (tile row 4 w + pt, column h3_row_pixel(i)).
    const int i = lane & 15, kb = lane >> 4;
    const char* const a0 = reinterpret_cast<const char*>(xin) + (4 * wv * WT_HX + h3_row_pixel(i)) * WT_PSB + h3_chunk_pos(kb, 0, 0);
    const char* const b0 = wbuf + lane * 16;                     // (the helpers' DMA laid the rows out for this: no permuted read here)
    int rb = 0;
    const float thr = a.relu ? 0.f : -__builtin_inff();          // ReLU without a branch: v < thr ? 0 : v (NaN stays NaN: torch.nn.ReLU)
#pragma unroll 1
    for (; item < nitems; item += gridDim.x) {
        const WTilePos q = wtile_pos(a, item);
        f32x4 mainv[2][2][4], corrv[2][2][4];                      // [pass g][tile e of the pair][tile row pt]
        h8 xh[4], xl[4];                                           // pixel fragments of the current K step: [pt]
        h8 wh[2][2], wl[2][2];                                     // weight fragments: [slot = K step][e]
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        // where this lane's results go (wepilogue of the wide kernel): float32 -- 128 g + 32 kb; split -- hi halves at 64 g + 16 kb
        const __amdgpu_buffer_rsrc_t ry = image_rsrc(a.y + (size_t)q.img * a.H * a.W * 64, a.H, a.W, pix, 0);
        int ln = lane;
        asm volatile("" : "+v"(ln));                             // (opaque: derived values are computed per item, not kept through the tap loop)
        const int gx = q.x0 + h3_row_pixel(ln & 15), okb = (YSPLIT ? 16 : 32) * (ln >> 4);
        const int pb0 = gx < a.W ? ((q.y0 + 4 * wv) * a.W + gx) * pix + okb : -256, pbs = gx < a.W ? a.W * pix : 0;
        f32x4 bs0[2];                                            // pass A's biases: held through pass B, where its pieces are finished
#define WP_LOAD_X(pt_, ap_, s2_)                                                                         \
        xh[pt_] = *reinterpret_cast<const h8*>((ap_) + (pt_) * (WT_HX * WT_PSB) + 32 * (s2_));           \
        xl[pt_] = *reinterpret_cast<const h8*>((ap_) + (pt_) * (WT_HX * WT_PSB) + 32 * (s2_) + 16);
        // weight fragments of (buffer, K step s2): unit (s2, e, part) at ((s2 * 2 + e) * 2 + part) KiB
#define WP_LOAD_W(slot_, bw_, s2_)                                                                       \
        _Pragma("unroll") for (int e_ = 0; e_ < 2; ++e_) {                                               \
            wh[slot_][e_] = *reinterpret_cast<const h8*>((bw_) + (s2_) * 4096 + 2048 * e_);              \
            wl[slot_][e_] = *reinterpret_cast<const h8*>((bw_) + (s2_) * 4096 + 2048 * e_ + 1024);       \
        }
        // the six MFMAs of pixel tile pt in pass g: main += w_hi x_hi; corr += w_lo x_hi; corr += w_hi x_lo (the narrow kernel's order per value)
#define WP_MFMA6(g_, slot_, pt_, z_)                                                                     \
        {                                                                                                \
            mainv[g_][0][pt_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[slot_][0], xh[pt_], (z_) ? zero4 : mainv[g_][0][pt_], 0, 0, 0);  \
            corrv[g_][0][pt_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[slot_][0], xh[pt_], (z_) ? zero4 : corrv[g_][0][pt_], 0, 0, 0);  \
            mainv[g_][1][pt_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[slot_][1], xh[pt_], (z_) ? zero4 : mainv[g_][1][pt_], 0, 0, 0);  \
            corrv[g_][1][pt_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[slot_][1], xh[pt_], (z_) ? zero4 : corrv[g_][1][pt_], 0, 0, 0);  \
            corrv[g_][0][pt_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[slot_][0], xl[pt_], corrv[g_][0][pt_], 0, 0, 0);                \
            corrv[g_][1][pt_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[slot_][1], xl[pt_], corrv[g_][1][pt_], 0, 0, 0);                \
        }
#define WP_FENCE __builtin_amdgcn_sched_barrier(0);
        // inside a pass-tap that also finishes a piece of pass A: per MFMA two vector instructions, the LDS reads where the plain pass-taps have
        // them (sched_group_barrier: 0x008 MFMA, 0x002 VALU, 0x100 DS read, 0x040 VMEM write)
#define WP_MIX6 _Pragma("unroll") for (int k_ = 0; k_ < 6; ++k_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, WP_MIX_VALU, 0); }
#pragma unroll
        for (int p = 0; p < WP_PT; ++p) {
            const int g = p / 9, tap = p % 9, ky = tap / 3, kx = tap - 3 * ky;
            const char* const ap = a0 + (ky * WT_HX + kx) * WT_PSB;               // input pixel of this tap
            const char* const bw = b0 + ((rb + p) & 3) * WP_WB;                   // this pass-tap's weight buffer and the next one's
            const char* const bw1 = b0 + ((rb + p + 1) & 3) * WP_WB;
            const bool z = tap == 0;                                              // compile-time: the pass's accumulators' first use (K step 0 only)
            const int tap1 = (p + 1) % 9, ky1 = tap1 / 3, kx1 = tap1 - 3 * ky1;
            const char* const ap1 = a0 + (ky1 * WT_HX + kx1) * WT_PSB;
            const bool mix = g == 1 && !(tap & 1) && tap < 8;                     // pass-taps 9, 11, 13, 15: pass A's tile row (tap / 2) is finished and stored
            if (p == 0) {
                // a new input tile: nothing of it may be read before B_0
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                WP_LOAD_X(0, ap, 0) WP_LOAD_X(1, ap, 0) WP_LOAD_X(2, ap, 0) WP_LOAD_X(3, ap, 0)
                WP_LOAD_W(0, bw, 0)
            } else {
                // its first fragments were requested during the pass-tap before (block P + 1 is in place since B_P): only the barrier
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if (p == 9) {                                                        // pass A's biases, for its pieces
#pragma unroll
                for (int e = 0; e < 2; ++e) bs0[e] = *reinterpret_cast<const f32x4*>(lbias + 8 * (ln >> 4) + 4 * e);
            }
            WP_FENCE
            if (!mix) {
                WP_LOAD_W(1, bw, 1)
                WP_FENCE
                // K step 0; a pixel fragment is dead after its six MFMAs: K step 1's takes its registers
                WP_MFMA6(g, 0, 0, z) WP_FENCE WP_LOAD_X(0, ap, 1) WP_FENCE
                WP_MFMA6(g, 0, 1, z) WP_FENCE WP_LOAD_X(1, ap, 1) WP_FENCE
                WP_MFMA6(g, 0, 2, z) WP_FENCE WP_LOAD_X(2, ap, 1) WP_FENCE
                WP_MFMA6(g, 0, 3, z) WP_FENCE WP_LOAD_X(3, ap, 1) WP_FENCE
                if (p + 1 < WP_PT) {
                    WP_LOAD_W(0, bw1, 0)
                    WP_FENCE
                    // K step 1; behind each pixel tile the fragment of the NEXT pass-tap's K step 0 (pass B re-reads the tile from tap 0)
                    WP_MFMA6(g, 1, 0, false) WP_FENCE WP_LOAD_X(0, ap1, 0) WP_FENCE
                    WP_MFMA6(g, 1, 1, false) WP_FENCE WP_LOAD_X(1, ap1, 0) WP_FENCE
                    WP_MFMA6(g, 1, 2, false) WP_FENCE WP_LOAD_X(2, ap1, 0) WP_FENCE
                    WP_MFMA6(g, 1, 3, false) WP_FENCE WP_LOAD_X(3, ap1, 0) WP_FENCE
                } else {
                    WP_MFMA6(g, 1, 0, false) WP_MFMA6(g, 1, 1, false) WP_MFMA6(g, 1, 2, false) WP_MFMA6(g, 1, 3, false)
                    WP_FENCE
                }
            } else {
                // the same reads and MFMAs with a piece of pass A's epilogue between them: ONE scheduling region, its order given below
                wp_piece<YSPLIT>(mainv[0], corrv[0], tap >> 1, bs0, thr, ry, pb0 + (tap >> 1) * pbs);
                WP_LOAD_W(1, bw, 1)
                WP_MFMA6(g, 0, 0, z) WP_LOAD_X(0, ap, 1)
                WP_MFMA6(g, 0, 1, z) WP_LOAD_X(1, ap, 1)
                WP_MFMA6(g, 0, 2, z) WP_LOAD_X(2, ap, 1)
                WP_MFMA6(g, 0, 3, z) WP_LOAD_X(3, ap, 1)
                WP_LOAD_W(0, bw1, 0)
                WP_MFMA6(g, 1, 0, false) WP_LOAD_X(0, ap1, 0)
                WP_MFMA6(g, 1, 1, false) WP_LOAD_X(1, ap1, 0)
                WP_MFMA6(g, 1, 2, false) WP_LOAD_X(2, ap1, 0)
                WP_MFMA6(g, 1, 3, false) WP_LOAD_X(3, ap1, 0)
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                WP_MIX6 __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                WP_MIX6 __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                WP_MIX6 __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                WP_MIX6 __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                WP_MIX6 __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                WP_MIX6 __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                WP_MIX6 __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                WP_MIX6 __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x040, 2, 0);
                WP_FENCE
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // E: this wave's reads of the input tile are back
        asm volatile("" ::: "memory");
        // pass B's half of the epilogue, while the helper waves hand the next input tile over
        {
            f32x4 bs1[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) bs1[e] = *reinterpret_cast<const f32x4*>(lbias + 32 + 8 * (ln >> 4) + 4 * e);
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) wp_piece<YSPLIT>(mainv[1], corrv[1], pt, bs1, thr, ry, pb0 + pt * pbs + (YSPLIT ? 64 : 128));
        }
        rb ^= 2;
#undef WP_LOAD_X
#undef WP_LOAD_W
#undef WP_MFMA6
    }
}

hipError_t launch_conv3x3_f16x3_pipe(hipStream_t s, const float* x, const float* w, const float* bias, float* y, int n, int H, int W, int relu, int fmt) {
    if (fmt & ~(CV_FMT_X | CV_FMT_Y)) return hipErrorInvalidValue;
    ConvArgs a;
    a.x = x; a.w = w; a.bias = bias; a.skip = nullptr; a.y = y; a.n = n; a.H = H; a.W = W; a.relu = relu; a.C = 64; a.fmt = fmt;
    a.tiles_x = (W + WT_TX - 1) / WT_TX; a.tiles_y = (H + WT_TY - 1) / WT_TY;
    const long long items = (long long)n * a.tiles_x * a.tiles_y;
    if (items <= 0 || items > 0x7fffffffLL) return hipErrorInvalidValue;
    if ((long long)H * W * 64 * 4 > 0x7fffffffLL) return hipErrorInvalidValue;          // one image must fit a signed 32-bit buffer offset
    const int cus = conv_compute_units();
    if (cus <= 0) return hipGetLastError();
    // persistent workgroups, ONE per compute unit; every workgroup's loop ends (item < nitems), all eight waves run the same trip counts
    const long long grid = items < cus ? items : cus;
    if (fmt & CV_FMT_Y) hipLaunchKernelGGL(k_conv3x3_h3p<true>, dim3((unsigned)grid), dim3(WT_THREADS), 0, s, a, (int)items);
    else hipLaunchKernelGGL(k_conv3x3_h3p<false>, dim3((unsigned)grid), dim3(WT_THREADS), 0, s, a, (int)items);
    return hipGetLastError();
}

}  // namespace pnp
