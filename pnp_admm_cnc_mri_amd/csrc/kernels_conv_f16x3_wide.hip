// The conv3x3 C -> C layer of kernels_conv_f16x3.hip (split-half arithmetic) with a 64 x 64 WAVE TILE and the work of a
// workgroup divided between two kinds of waves: "wide".  Round 6; models/basicblock.py:63-100, network_ffdnet.py:58-73, network_unet.py:95-136.
//
// Why.  The round-5 counters of k_conv3x3_c64_h3<1> (profiles/pmc_conv_f16x3_r05.txt) show the matrix pipe busy 0.55 of the cycles on
// random and on zero data alike: the rest is not power but what every wave does besides its MFMAs -- per tap of 48 MFMAs a wave also issues
// four LDS-DMAs, two prefetch loads, 24 ds_read_b128, counted waits and a barrier, and per item the split of the next input tile, its LDS
// writes and an epilogue staged through LDS.  Here
//   * a wave owns 64 pixels (4 tile rows of 16) x 64 output channels: 96 MFMAs per tap for 32 ds_read_b128 (one read per three MFMAs
//     instead of one per two), and a tap's 16 KiB of weights serve 256 pixels instead of 128;
//   * ONE workgroup of 512 threads per compute unit: waves 0-3 COMPUTE -- their instruction stream inside a tap is LDS reads and MFMAs and
//     nothing else --, waves 4-7 (one beside each compute wave on its SIMD) are HELPERS: they issue every LDS-DMA of the weight stream, fetch
//     the next input tile into registers piece by piece, split it and write it into LDS between two chunks, and count the waits; what
//     they do overlaps the partner's MFMAs instead of interrupting them;
//   * weights in FOUR buffers, three taps ahead: at the barrier that opens tap T the weights of taps T and T + 1 have landed, so a compute
//     wave reads its first fragments of tap T + 1 before that tap's barrier -- after a barrier the next instruction is an MFMA;
//   * the matrix product is taken TRANSPOSED (A = weights: M = output channels, B = pixels): an accumulator quad of a lane is then four
//     consecutive channels of ONE pixel and leaves as a 16-byte store straight from the registers -- no LDS staging, no wave barrier, no
//     second workgroup barrier per item.  For the split activation format two v_permlane16_swap per store pair the lane rows up so that
//     each lane still stores 16 bytes (eight hi halves or eight lo halves).
// The arithmetic is the narrow kernel's, instruction for instruction per output value (same products, same order of accumulation):
// tests/test_gpu_conv.py holds the two kernels BIT-equal.  16 x 16 pixel tiles: 86 KiB of input tile + 64 KiB of weights in LDS.
// Measured (profiles/conv_f16x3_wide_r06.txt): 3.39e6 cycles per [640, 64, 128, 128] launch against the narrow kernel's 3.99e6, matrix pipe busy 0.65
// against 0.55; per item of 13.8 k MFMA cycles the taps take 15.4 k, the epilogue 4.6 k, the barriers 1.2 k.
#include "f16x3_wide_common.h"
#include <type_traits>

namespace pnp {

// The epilogue of an item, straight from the accumulators: y = relu?(acc + bias + skip), ONE 16-byte store per lane and (ct, pt).
// SKIP / KSPLIT / YSPLIT: a skip tensor is added / it is in the split activation format / y is written in it (f16x3_common.h) -- compile-time,
// one instance each: sixteen stores behind uniform branches would put a dozen branches per store on the wave's critical path.
// The two orders of the sum are the narrow kernel's (store_rows32 / store_rows32_fmt): all-float32 tensors acc + (skip + bias), any split
// tensor (acc + bias) + skip.
template <bool SKIP, bool KSPLIT, bool YSPLIT>
__device__ __forceinline__ void wepilogue(const ConvArgs& a, const WTilePos& q, const f32x4 (&mainv)[4][4], const f32x4 (&corrv)[4][4],
                                          int wv, int lane, const int pix, const int cb, const float* lbias) {
    const __amdgpu_buffer_rsrc_t ry = image_rsrc(a.y + (size_t)q.img * a.H * a.W * (pix >> 2), a.H, a.W, pix, 64 * cb);
    const __amdgpu_buffer_rsrc_t rk = image_rsrc((SKIP ? a.skip : a.y) + (size_t)q.img * a.H * a.W * (pix >> 2), a.H, a.W, pix, 64 * cb);
    // (opaque copy of the lane number: what is derived from it -- and the biases -- is computed HERE, per item; hoisted out of the item loop as
    // the loop invariants they are, these values would live through the tap loop, which has no register to spare)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int i = ln & 15, kb = ln >> 4;
    constexpr bool FMT = KSPLIT || YSPLIT;
    const int gx = q.x0 + h3_row_pixel(i);
    // byte offsets of this lane's eight channels 32 g + 8 kb .. inside the pixel's block of 64: float32 -- 128 g + 32 kb (+ 16: the second four);
    // split -- hi halves at 64 g + 16 kb, lo halves 128 bytes behind them
    const int of32 = 32 * kb, osp = 16 * kb;
    // ReLU without a branch: v < thr ? 0 : v with thr = 0, or -inf (never true; NaN stays NaN either way: torch.nn.ReLU)
    const float thr = a.relu ? 0.f : -__builtin_inff();
    // byte offset of this lane's pixel in tile row pt; rows below the image: beyond the buffer's range by themselves; columns right of
    // it: such an offset
#ifdef WT_ABL_OOB                                                 // timing-only ablation (results wrong by design): every store out of range, dropped by the address unit
    const int pb0 = -256, pbs = 0;
#else
    const int pb0 = gx < a.W ? ((q.y0 + 4 * wv) * a.W + gx) * pix : -256, pbs = gx < a.W ? a.W * pix : 0;
#endif
#define WT_PB(pt_) (pb0 + (pt_) * pbs)
#ifdef WT_ABL_NOST                                                // timing-only ablation: the values are computed, no store is issued
#define WT_STORE(o_, off_) asm volatile("" :: "v"(o_))
#else
// (cache-policy bits on these stores -- sc0, nt, both -- measured within +-1 % of none: profiles/experiments/ab_store_cache_policy_wide_r06.txt)
#define WT_STORE(o_, off_) __builtin_amdgcn_raw_buffer_store_b128(o_, ry, off_, 0, 0)
#endif
    f32x4 bs[2][2];                                              // [g][e]: this lane's biases, from LDS (the helpers put them there)
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int e = 0; e < 2; ++e) bs[g][e] = *reinterpret_cast<const f32x4*>(lbias + 32 * g + 8 * kb + 4 * e);
    u32x4v kq[SKIP ? 4 : 1][2][2];                               // [pt][g][float32: e / split: hi, lo]
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        if (SKIP && !(pt & 1)) {                                 // the eight requests of two tile rows at once: two memory round trips per item
#pragma unroll                                                   // (all sixteen at once: 64 registers more than the accumulators leave)
            for (int p2 = pt; p2 < pt + 2; ++p2)
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        kq[p2][g][e] = __builtin_amdgcn_raw_buffer_load_b128(rk, WT_PB(p2) + (KSPLIT ? 64 * g + osp + 128 * e : 128 * g + of32 + 16 * e), 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f32x4 v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ct = 2 * g + e;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[e][r] = fmaf(corrv[ct][pt][r], H3_RSCALE, mainv[ct][pt][r]);
            }
            if (SKIP) {
                f32x4 k[2];
                if (KSPLIT) {
                    const h8 kh = __builtin_bit_cast(h8, kq[pt][g][0]), kl = __builtin_bit_cast(h8, kq[pt][g][1]);
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int r = 0; r < 4; ++r) k[e][r] = unsplit(kh[4 * e + r], kl[4 * e + r]);
                } else {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const u32x4v k_ = kq[pt][g][e];
                        k[e] = f32x4{__uint_as_float(k_.x), __uint_as_float(k_.y), __uint_as_float(k_.z), __uint_as_float(k_.w)};
                    }
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) v[e] = FMT ? (v[e] + bs[g][e]) + k[e] : v[e] + (k[e] + bs[g][e]);
            } else {
#pragma unroll
                for (int e = 0; e < 2; ++e) v[e] += bs[g][e];
            }
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[e][r] = v[e][r] < thr ? 0.f : v[e][r];
            if (YSPLIT) {
                h4 h0, l0, h1, l1;
                split4(v[0], h0, l0);
                split4(v[1], h1, l1);
                const h8 hi = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]}, lo = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                WT_STORE(__builtin_bit_cast(u32x4v, hi), WT_PB(pt) + 64 * g + osp);
                WT_STORE(__builtin_bit_cast(u32x4v, lo), WT_PB(pt) + 64 * g + osp + 128);
            } else {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const u32x4v o = {__float_as_uint(v[e][0]), __float_as_uint(v[e][1]), __float_as_uint(v[e][2]), __float_as_uint(v[e][3])};
                    WT_STORE(o, WT_PB(pt) + 128 * g + of32 + 16 * e);
                }
            }
        }
    }
#undef WT_PB
#undef WT_STORE
}

#define WT_PIECE_LOADS(k_) ((k_) > 5 ? 0 : (k_) * WT_PIECE >= WT_XU ? 0 : ((k_) + 1) * WT_PIECE <= WT_XU ? WT_PIECE : WT_XU - (k_) * WT_PIECE)

#ifdef H3W_PROF
// diagnostic build (profiles/variants.sh build kernels_conv_f16x3_wide.hip h3wprof "-DH3W_PROF"): shader-clock sums per phase, compute wave 0 of every
// workgroup.  The stamp (cdna_hip_programming.md section 7): s_memtime + lgkmcnt(0) as ONE statement -- it drains the LDS reads the tap loop keeps in
// flight across a barrier, so the instrumented build is a little slower than the real one (stamps sit at barriers only).
__device__ unsigned g_h3wprof[1024 * 8];
#define WT_STAMP(k) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                      __builtin_amdgcn_sched_barrier(0); psum[k] += (unsigned)(t_ - tlast); tlast = t_; }
#else
#define WT_STAMP(k)
#endif

__global__ __launch_bounds__(WT_THREADS, 2) void k_conv3x3_h3w(ConvArgs a, int nitems) {
    // ONE LDS array: the input tile, then the three weight buffers
    __shared__ __attribute__((aligned(16))) float lds[WT_XIN + WT_NBUF * H3_TAP16 * 4 + 64];
    float* const xin = lds;
    float* const lbias = lds + WT_XIN + WT_NBUF * H3_TAP16 * 4;   // the 64 biases of this workgroup's block of output channels (its cb never changes)
    f32x4 (*const wbuf)[H3_TAP16] = reinterpret_cast<f32x4 (*)[H3_TAP16]>(lds + WT_XIN);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // an ITEM is (tile, block cb of 64 output channels); item = tile * NC + cb, gridDim.x a multiple of NC: a workgroup keeps its cb, its
    // weight stream -- [cb][chunk][tap] blocks of 16 KiB -- is periodic in 9 NC taps (kernels_conv_f16x3.hip)
    const int NC = a.C >> 6, pix = a.C * 4, period = 9 * NC;
    int item = blockIdx.x;
    if (item >= nitems) return;                                  // (uniform over the workgroup)
    const int cb = item % NC;
    const bool in_split = (a.fmt & CV_FMT_X) != 0;

    // BARRIER PLAN -- identical for the eight waves: per chunk of 64 input channels nine barriers B_tap, one in front of every tap,
    // and one barrier E behind the last tap.  At B_T (global tap counter T): the weights of taps T and T + 1 have landed in buffers T % 4,
    // (T + 1) % 4 (each helper waited for its own DMAs before arriving); every compute wave is done with tap T - 1, so buffer (T + 3) % 4 may
    // be overwritten; at B_0 of a chunk the input tile is in place.  At E every compute wave is done with the input tile.
    if (wv >= 4) {
        // ------------------------------------------------ HELPER waves (4..7) ------------------------------------------------
        const int h = tid - WT_HTHREADS, hw = wv - 4;
        WStaging st;
        wstaging_init(a, h, st, pix);
        f32x4 xpre[WT_XU];
        {
            const WFetch f0 = wfetch_begin(a, wtile_pos(a, item / NC), h, pix, 0);
            wfetch_piece<0, WT_XU>(f0, st, xpre);
        }
        // A tap's 16 KiB of weights: global memory -> LDS by LDS-DMA (`buffer_load_dwordx4 ... offen lds`, the tap in the SCALAR offset): helper
        // wave w copies units w * 64 + lane + 256 j of the 1024.  FOUR buffers, global tap T in buffer T % 4: in tap T the block of tap T + 3
        // is requested into the buffer tap T - 1 has left, and the wait at the end of tap T leaves that request (and the input piece behind
        // it) in flight -- what it retires is the block of tap T + 2, requested a whole tap earlier.  At B_(T+1) the weights of taps T + 1 and
        // T + 2 are in place: the compute waves may read tap T + 2's first fragments before B_(T+2).  (Three buffers with the wait one group
        // later -- this kernel's first version -- let that read race the DMA; three buffers with the wait in the requesting tap put a DMA's
        // latency on the barrier: profiles/conv_f16x3_wide_r06.txt.)
        // The DMA also PERMUTES the rows (f16x3_wide_common.h): LDS unit u = (s2 * 4 + ct) * 2 + part is lane (i, kb)'s fragment of channel tile
        // ct = 2 g + e, whose row i is output channel 32 g + 8 (i >> 2) + 4 e + (i & 3) -- fetched from packed fragment nt = 2 g + (i >> 3) by the
        // per-lane GLOBAL address (wt_wrow_offset) -- so the compute waves read at lane * 16 with no bank conflict and still hold eight
        // consecutive channels per tile pair.  Helper wave w copies units w + 4 j; their positions in the packed tap go in the SCALAR offset.
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, 9 * a.C * a.C * 4, 0x00020000);
        const int wvoff = wt_wrow_offset(lane & 15, lane >> 4), wbase = cb * period * (H3_TAP16 * 16);
        f32x4* const wb0 = &wbuf[0][hw * 64];
        int usrc[4];                                             // unit u = hw + 4 j -> byte offset of (s2, g, part, e) in the packed tap
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = hw + 4 * j, part = u & 1, ct = (u >> 1) & 3, s2 = u >> 3;
            usrc[j] = __builtin_amdgcn_readfirstlane(1024 * ((s2 * 4 + (ct & 2)) * 2 + part) + 64 * (ct & 1));
        }
#define WT_DMA(rot_, t_)                                                                                                      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                          \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(wb0 + (rot_) * H3_TAP16 + 256 * j), 16, wvoff, \
                                                     wbase + (t_) * (H3_TAP16 * 16) + usrc[j], 0, 0);
        int t2 = 0;                                              // stream position of the NEXT tap to request
        WT_DMA(0, t2) t2 = t2 + 1 == period ? 0 : t2 + 1;
        WT_DMA(1, t2) t2 = t2 + 1 == period ? 0 : t2 + 1;
        WT_DMA(2, t2) t2 = t2 + 1 == period ? 0 : t2 + 1;
        int rot = 3;                                             // buffer of the next request = (global tap + 3) % 4
        wput_input(xin, h, xpre, in_split);
        {   // the biases, read by the compute waves' epilogues from LDS instead of a memory round trip per item (no bias: a descriptor of zero bytes returns zeros)
            const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias ? a.bias : a.w), 0, a.bias ? a.C * 4 : 0, 0x00020000);
            const float bv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, (64 * cb + (h & 63)) * 4, 0, 0));
            if (h < 64) lbias[h] = bv;
        }
        WT_WAIT_VM(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // raw barriers: this wave's LDS writes are done before it arrives
#pragma unroll 1
        for (; item < nitems; item += gridDim.x) {
            const WTilePos q = wtile_pos(a, item / NC);
#pragma unroll 1
            for (int cc = 0; cc < NC; ++cc) {
                // the input tile that follows this one: the tile's next 64 input channels, or the first 64 of the next item
                const bool last = cc + 1 == NC;
                const bool more = !last || item + (int)gridDim.x < nitems;
                const WFetch nx = wfetch_begin(a, last && more ? wtile_pos(a, (item + gridDim.x) / NC) : q, h, pix, last ? 0 : 64 * (cc + 1), more);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    __builtin_amdgcn_s_barrier();                 // B_tap
                    asm volatile("" ::: "memory");
                    WT_DMA(rot, t2)                               // the weights of tap T + 3 into the buffer tap T - 1 has left
                    t2 = t2 + 1 == period ? 0 : t2 + 1;
                    rot = (rot + 1) & 3;
                    // the counted wait below is right only if the four DMAs are OLDER than the piece: pin the order (tools/isa_scan.py checks it)
                    __builtin_amdgcn_sched_barrier(0);
                    if (tap == 0) wfetch_piece<0 * WT_PIECE, 1 * WT_PIECE>(nx, st, xpre);
                    if (tap == 1) wfetch_piece<1 * WT_PIECE, 2 * WT_PIECE>(nx, st, xpre);
                    if (tap == 2) wfetch_piece<2 * WT_PIECE, 3 * WT_PIECE>(nx, st, xpre);
                    if (tap == 3) wfetch_piece<3 * WT_PIECE, 4 * WT_PIECE>(nx, st, xpre);
                    if (tap == 4) wfetch_piece<4 * WT_PIECE, 5 * WT_PIECE>(nx, st, xpre);
                    if (tap == 5) wfetch_piece<5 * WT_PIECE, 6 * WT_PIECE>(nx, st, xpre);
                    __builtin_amdgcn_sched_barrier(0);
                    // everything but this tap's four DMAs and its piece has completed
                    if (tap == 0) { WT_WAIT_VM(4 + WT_PIECE_LOADS(0)); }
                    if (tap == 1) { WT_WAIT_VM(4 + WT_PIECE_LOADS(1)); }
                    if (tap == 2) { WT_WAIT_VM(4 + WT_PIECE_LOADS(2)); }
                    if (tap == 3) { WT_WAIT_VM(4 + WT_PIECE_LOADS(3)); }
                    if (tap == 4) { WT_WAIT_VM(4 + WT_PIECE_LOADS(4)); }
                    if (tap == 5) { WT_WAIT_VM(4 + WT_PIECE_LOADS(5)); }
                    if (tap >= 6) { WT_WAIT_VM(4); }
                }
                __builtin_amdgcn_s_barrier();                     // E: the compute waves are done with the input tile
                asm volatile("" ::: "memory");
                // (every piece is older than the DMAs of taps 6..8, which the waits above retired up to the last four: xpre is complete)
                if (more) wput_input(xin, h, xpre, in_split);     // published by B_0 of the next chunk
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        // no wave ends with an LDS-DMA in flight (its LDS may belong to the next workgroup by the time the data lands)
        WT_WAIT_VM(0);
#undef WT_DMA
        return;
    }

    // ------------------------------------------------ COMPUTE waves ------------------------------------------------
    // v_mfma_f32_16x16x32_f16, D = A B with A = the weights (row = output channel) and B = the pixels (column = pixel): lane (i, kb)
    // supplies A[channel i][k = 8 kb ..] and B[k = 8 kb ..][pixel i]; the wave's 64 pixels are four N tiles = its four tile rows (i =
    // the pixel's column through h3_row_pixel), its 64 output channels four M tiles.  Accumulator reg r of lane (i, kb), tile (ct, pt) =
    // channel 16 ct + 4 kb + r of pixel (tile row 4 w + pt, column h3_row_pixel(i)).
    const int i = lane & 15, kb = lane >> 4;
    const char* const a0 = reinterpret_cast<const char*>(xin) + (4 * wv * WT_HX + h3_row_pixel(i)) * WT_PSB + h3_chunk_pos(kb, 0, 0);
    const char* const b0 = reinterpret_cast<const char*>(&wbuf[0][0]) + lane * 16;                  // (the helpers' DMA permuted the rows: wepilogue's note)
    int rot = 0;                                                 // buffer of the current tap = global tap % 4 (uniform)
#ifdef H3W_PROF
    unsigned psum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_readcyclecounter();
#endif
#pragma unroll 1
    for (; item < nitems; item += gridDim.x) {
        const WTilePos q = wtile_pos(a, item / NC);
        f32x4 mainv[4][4], corrv[4][4];                            // [M tile ct = 16 output channels][N tile pt = tile row of the wave]
        auto chunk = [&](const int cc, auto first_tag) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            h8 xh[4], xl[4];                                       // pixel fragments of the current K step: [pt]
            h8 wh[2][2], wl[2][2];                                 // weight fragments: [slot][ct of the pair]
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            // pixel fragment (tap, K step s2, tile row pt): + 32 s2: K step, + 16: the lo halves (f16x3_common.h)
#define WT_LOAD_X(pt_, ap_, s2_)                                                                         \
            xh[pt_] = *reinterpret_cast<const h8*>((ap_) + (pt_) * (WT_HX * WT_PSB) + 32 * (s2_));       \
            xl[pt_] = *reinterpret_cast<const h8*>((ap_) + (pt_) * (WT_HX * WT_PSB) + 32 * (s2_) + 16);
            // weight fragments of (buffer, K step s2, channel-tile pair cp): LDS unit (s2 * 4 + ct) * 2 + part, 1 KiB each
#define WT_LOAD_W(slot_, bw_, s2_, cp_)                                                                  \
            _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_) {                                           \
                wh[slot_][c_] = *reinterpret_cast<const h8*>((bw_) + 1024 * ((((s2_) * 4 + 2 * (cp_) + c_) * 2)));      \
                wl[slot_][c_] = *reinterpret_cast<const h8*>((bw_) + 1024 * ((((s2_) * 4 + 2 * (cp_) + c_) * 2) + 1));  \
            }
            // the six MFMAs of (pixel tile pt) x (channel-tile pair cp): main += w_hi x_hi; corr += w_lo x_hi; corr += w_hi x_lo -- the order
            // of the narrow kernel (x_hi w_hi; x_hi w_lo; x_lo w_hi), the two dependent corr updates two instructions apart
#define WT_MFMA6(slot_, cp_, pt_, z_)                                                                    \
            {                                                                                            \
                const int c0_ = 2 * (cp_), c1_ = 2 * (cp_) + 1;                                          \
                mainv[c0_][pt_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[slot_][0], xh[pt_], (z_) ? zero4 : mainv[c0_][pt_], 0, 0, 0);  \
                corrv[c0_][pt_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[slot_][0], xh[pt_], (z_) ? zero4 : corrv[c0_][pt_], 0, 0, 0);  \
                mainv[c1_][pt_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[slot_][1], xh[pt_], (z_) ? zero4 : mainv[c1_][pt_], 0, 0, 0);  \
                corrv[c1_][pt_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[slot_][1], xh[pt_], (z_) ? zero4 : corrv[c1_][pt_], 0, 0, 0);  \
                corrv[c0_][pt_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[slot_][0], xl[pt_], corrv[c0_][pt_], 0, 0, 0);                \
                corrv[c1_][pt_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[slot_][1], xl[pt_], corrv[c1_][pt_], 0, 0, 0);                \
            }
#define WT_FENCE __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - 3 * ky;
                const char* const ap = a0 + (ky * WT_HX + kx) * WT_PSB;           // input pixel of this tap
                // this tap's weight buffer and the next one's: ONE vector add per tap each (the buffer rotates through four, a chunk has nine taps)
                const char* const bw = b0 + rot * (H3_TAP16 * 16);
                const char* const bw1 = b0 + ((rot + 1) & 3) * (H3_TAP16 * 16);
                rot = (rot + 1) & 3;
                const bool z = FIRST && tap == 0;                               // compile-time: the accumulators' first use (K step 0 only)
                if (tap == 0) {
                    // a new input tile: nothing of it may be read before B_0 (the weights of this tap landed long ago, but keep it simple)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    WT_STAMP(0)                                              // (0: what lies between two chunks besides the barriers: epilogue excluded)
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    WT_STAMP(1)                                              // 1: waiting at B_0 (the input waves' hand-over)
                    WT_LOAD_X(0, ap, 0) WT_LOAD_X(1, ap, 0) WT_LOAD_X(2, ap, 0) WT_LOAD_X(3, ap, 0)
                    WT_LOAD_W(0, bw, 0, 0)
                } else {
                    // its first fragments were requested during the tap before (weights T + 1 are in place since B_T): only the barrier
                    WT_STAMP(2)                                              // 2: the taps (LDS reads + MFMAs)
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    WT_STAMP(3)                                              // 3: waiting at B_1..8
                }
                WT_FENCE
                // half step 0: K step 0, channel tiles 0, 1
                WT_LOAD_W(1, bw, 0, 1)
                WT_FENCE
                WT_MFMA6(0, 0, 0, z) WT_MFMA6(0, 0, 1, z) WT_MFMA6(0, 0, 2, z) WT_MFMA6(0, 0, 3, z)
                WT_FENCE
                // half step 1: K step 0, channel tiles 2, 3 -- a pixel fragment is dead after its six MFMAs: K step 1's takes its registers
                WT_LOAD_W(0, bw, 1, 0)
                WT_FENCE
                WT_MFMA6(1, 1, 0, z) WT_FENCE WT_LOAD_X(0, ap, 1) WT_FENCE
                WT_MFMA6(1, 1, 1, z) WT_FENCE WT_LOAD_X(1, ap, 1) WT_FENCE
                WT_MFMA6(1, 1, 2, z) WT_FENCE WT_LOAD_X(2, ap, 1) WT_FENCE
                WT_MFMA6(1, 1, 3, z) WT_FENCE WT_LOAD_X(3, ap, 1) WT_FENCE
                // half step 2: K step 1, channel tiles 0, 1
                WT_LOAD_W(1, bw, 1, 1)
                WT_FENCE
                WT_MFMA6(0, 0, 0, false) WT_MFMA6(0, 0, 1, false) WT_MFMA6(0, 0, 2, false) WT_MFMA6(0, 0, 3, false)
                WT_FENCE
                // half step 3: K step 1, channel tiles 2, 3; behind each pixel tile the fragment of the NEXT tap's K step 0
                if (tap + 1 < 9) {
                    const int ky1 = (tap + 1) / 3, kx1 = tap + 1 - 3 * ky1;
                    const char* const ap1 = a0 + (ky1 * WT_HX + kx1) * WT_PSB;
                    WT_LOAD_W(0, bw1, 0, 0)
                    WT_FENCE
                    WT_MFMA6(1, 1, 0, false) WT_FENCE WT_LOAD_X(0, ap1, 0) WT_FENCE
                    WT_MFMA6(1, 1, 1, false) WT_FENCE WT_LOAD_X(1, ap1, 0) WT_FENCE
                    WT_MFMA6(1, 1, 2, false) WT_FENCE WT_LOAD_X(2, ap1, 0) WT_FENCE
                    WT_MFMA6(1, 1, 3, false) WT_FENCE WT_LOAD_X(3, ap1, 0) WT_FENCE
                } else {
                    WT_MFMA6(1, 1, 0, false) WT_MFMA6(1, 1, 1, false) WT_MFMA6(1, 1, 2, false) WT_MFMA6(1, 1, 3, false)
                    WT_FENCE
                }
            }
#undef WT_LOAD_X
#undef WT_LOAD_W
#undef WT_MFMA6
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            WT_STAMP(2)
            __builtin_amdgcn_s_barrier();                            // E: this wave's reads of the input tile are back
            asm volatile("" ::: "memory");
            WT_STAMP(4)                                                  // 4: waiting at E
            (void)cc;
        };
        chunk(0, std::true_type{});
#pragma unroll 1
        for (int cc = 1; cc < NC; ++cc) chunk(cc, std::false_type{});

        // ---- epilogue, straight from the accumulators (wepilogue above), one instance per combination of formats ----
        switch ((a.skip ? 1 : 0) | ((a.fmt & CV_FMT_SKIP) ? 2 : 0) | ((a.fmt & CV_FMT_Y) ? 4 : 0)) {
            case 0: wepilogue<false, false, false>(a, q, mainv, corrv, wv, lane, pix, cb, lbias); break;
            case 1: wepilogue<true, false, false>(a, q, mainv, corrv, wv, lane, pix, cb, lbias); break;
            case 3: wepilogue<true, true, false>(a, q, mainv, corrv, wv, lane, pix, cb, lbias); break;
            case 4: case 6: wepilogue<false, false, true>(a, q, mainv, corrv, wv, lane, pix, cb, lbias); break;
            case 5: wepilogue<true, false, true>(a, q, mainv, corrv, wv, lane, pix, cb, lbias); break;
            case 7: wepilogue<true, true, true>(a, q, mainv, corrv, wv, lane, pix, cb, lbias); break;
            default: wepilogue<false, false, false>(a, q, mainv, corrv, wv, lane, pix, cb, lbias); break;      // (2: a skip format without a skip tensor)
        }
        WT_STAMP(5)                                                      // 5: the epilogue (issue side: the stores drain behind it)
    }
#ifdef H3W_PROF
    if (tid == 0 && blockIdx.x < 1024) for (int k = 0; k < 8; ++k) g_h3wprof[blockIdx.x * 8 + k] = psum[k];
#endif
}

#ifdef H3W_PROF
extern "C" int pnp_conv_h3w_prof_read(unsigned* out /* [1024][8] */) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_h3wprof), sizeof(unsigned) * 1024 * 8);
}
#endif

hipError_t launch_conv3x3_f16x3_wide(hipStream_t s, const float* x, const float* w, const float* bias, const float* skip, float* y,
                                     int n, int C, int H, int W, int relu, int fmt) {
    if (C < 64 || C > 1024 || (C & 63) || (fmt & ~(CV_FMT_X | CV_FMT_SKIP | CV_FMT_Y))) return hipErrorInvalidValue;
    ConvArgs a;
    a.x = x; a.w = w; a.bias = bias; a.skip = skip; a.y = y; a.n = n; a.H = H; a.W = W; a.relu = relu; a.C = C; a.fmt = fmt;
    a.tiles_x = (W + WT_TX - 1) / WT_TX; a.tiles_y = (H + WT_TY - 1) / WT_TY;
    const int NC = C >> 6;
    const long long items = (long long)n * a.tiles_x * a.tiles_y * NC;
    if (items <= 0 || items > 0x7fffffffLL) return hipErrorInvalidValue;
    if ((long long)H * W * C * 4 > 0x7fffffffLL) return hipErrorInvalidValue;           // one image must fit a signed 32-bit buffer offset
    const int cus = conv_compute_units();
    if (cus <= 0) return hipGetLastError();
    // persistent workgroups, ONE per compute unit, a multiple of NC of them (a workgroup keeps its block of output channels); every
    // workgroup's loop ends: item < nitems, and all eight waves of a workgroup run the same trip counts (the barrier plan)
    long long grid = cus;
    grid -= grid % NC;
    if (grid < NC) grid = NC;
    if (items < grid) grid = items;                               // items = tiles * NC: a multiple of NC as well
    hipLaunchKernelGGL(k_conv3x3_h3w, dim3((unsigned)grid), dim3(WT_THREADS), 0, s, a, (int)items);
    return hipGetLastError();
}

}  // namespace pnp
