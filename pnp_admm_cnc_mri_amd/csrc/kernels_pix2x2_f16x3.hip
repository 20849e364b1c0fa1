// DRUNet's scale changes in the split-half ("f16x3") arithmetic of kernels_conv_f16x3.hip:
//
//     down   torch.nn.Conv2d(C, 2C, 2, 2, 0, bias=False)            models/network_unet.py:95-99,  models/basicblock.py:415-421
//     up     torch.nn.ConvTranspose2d(C, C/2, 2, 2, 0, bias=False)  models/network_unet.py:103-107, models/basicblock.py:439-445
//
// Neither has a halo: both are plain matrix products over pixels,
//     down   y[(oy, ox)][co]               = sum_{dy, dx, ci} x[(2 oy + dy, 2 ox + dx)][ci] W[co][ci][dy][dx]      K = 4 C,  N = 2 C
//     up     y[(2 iy + dy, 2 ix + dx)][co] = sum_ci x[(iy, ix)][ci] W[ci][co][dy][dx]                              K = C,    N = 4 (C / 2)
// so one kernel serves the two: a workgroup item is 8 x 16 pixels of the TILE GRID (output pixels for `down`, input pixels for `up`) x
// one block of 64 matrix columns; its K loop runs over chunks of 64 input channels.  Per chunk the 128 pixels x 64 channels of A are
// loaded two chunks ahead into registers, split into half pairs on the way into LDS (pixel = [64 hi][64 lo] + 16 bytes, the conv
// kernel's layout, so an operand fragment is one ds_read_b128), the chunk's 64 x 64 weights arrive pre-split by LDS-DMA into the
// other of two 16 KiB buffers; 48 v_mfma_f32_16x16x32_f16 per wave and chunk (2 K steps x 2 M tiles x 4 N tiles x 3 products).
// Unlike the 3 x 3 kernel every chunk brings a new A tile: two barriers per chunk, and the big instances (64 <-> 128 channels at full
// resolution) are bound by their memory traffic, not by the matrix pipe -- which is all this kernel has to reach: the six layers are
// 2.3 % of DRUNet's arithmetic and were 8 % of its time on MIOpen.
// `x2`: an optional second input ADDED to x while staging -- the U-Net's skip additions `m_up(x + x_skip)` (models/network_unet.py:
// 131-133) ride on the transposed convolution that consumes the sum; the sum itself never goes to memory.
#include "f16x3_common.h"

namespace pnp {

struct Pix2Args {
    const float* x;      // [n][Hin][Win][Cin]
    const float* x2;     // null, or a tensor of x's shape added to it
    const float* w;      // packed: blocks [cb][kc] of 16 KiB (k_pix2_pack_w)
    float* y;            // [n][Hout][Wout][Cout]
    int n, Hin, Win, Cin, Hout, Wout, Cout;
    int GH, GW, tiles_x, tiles_y;      // the tile grid (down: Hout x Wout; up: Hin x Win) and its 8 x 16 tiling
    int KC, NB;                         // chunks of 64 along K, blocks of 64 matrix columns
};
constexpr int P2_ROWS = 8, P2_COLS = 16;
constexpr int P2_XIN = P2_ROWS * P2_COLS * CV_PS;          // floats of the A tile = of the epilogue's staging area (128 x 68)

template <bool UP, bool X2>
__global__ __launch_bounds__(CV_THREADS, 2) void k_pix2x2_h3(Pix2Args a, int nitems) {
    __shared__ __attribute__((aligned(16))) float lds[P2_XIN + 2 * H3_TAP16 * 4];      // one array: A tile, then the two weight buffers
    float* const xin = lds;
    f32x4 (*const wbuf)[H3_TAP16] = reinterpret_cast<f32x4 (*)[H3_TAP16]>(lds + P2_XIN);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kb = lane >> 4;
    const int NB = a.NB, KC = a.KC, ncc = a.Cin >> 6;
    int item = blockIdx.x;
    if (item >= nitems) return;
    const int cb = item % NB;                                   // gridDim.x is a multiple of NB: a workgroup keeps its block of columns
    const int pixA = a.Cin * 4, pixO = a.Cout * 4;
    const int per_img = a.tiles_x * a.tiles_y;
    // staging role of this thread: tile column sc, channels 4 sq .. of the chunk, the eight tile rows u
    const int sc = tid >> 4, sq = tid & 15;

    // A is requested TWO chunks ahead (a chunk's 48 MFMAs per wave are 0.4 us, a memory round trip under load several times that): two
    // register sets, chunk kc lives in set kc & 1 (KC is even).  The second tensor (X2) has ONE set, requested one chunk ahead: its
    // values must stay apart from x's until the operand is split (an addition at load time would wait for the loads on the spot), and
    // two more sets do not fit 256 registers.  `any` = false: a descriptor of zero bytes -- the loads are still ISSUED (the counted wait
    // below relies on their number) but reach no memory.
    f32x4 areg[2][P2_ROWS], breg[X2 ? P2_ROWS : 1];
    auto load_t = [&](const float* base, f32x4* dst, int it_, int kc_, const bool any) __attribute__((always_inline)) {
        const int t = it_ / NB, img = t / per_img, trem = t - img * per_img, ty = trem / a.tiles_x;
        const int gy0 = ty * P2_ROWS, gx = (trem - ty * a.tiles_x) * P2_COLS + sc;
        int dy = 0, dx = 0, cc = kc_;
        if (!UP) { const int q = kc_ / ncc; cc = kc_ - q * ncc; dy = q >> 1; dx = q & 1; }
        const size_t img_off = (size_t)img * a.Hin * a.Win * a.Cin;
        const unsigned bytes = any ? (unsigned)a.Hin * (unsigned)a.Win * (unsigned)pixA : 0u;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + img_off), 0, (int)bytes, 0x00020000);
        const int ix = UP ? gx : 2 * gx + dx;
        const int col_off = ix * pixA + (64 * cc + 4 * sq) * 4;
#pragma unroll
        for (int u = 0; u < P2_ROWS; ++u) {
            const int gy = gy0 + u, iy = UP ? gy : 2 * gy + dy;
            const int off = (gy < a.GH && gx < a.GW) ? iy * a.Win * pixA + col_off : -16;     // outside the grid: out of range, zeros
            const u32x4v w0 = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
            dst[u] = f32x4{__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z), __uint_as_float(w0.w)};
        }
    };
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, NB * KC * (H3_TAP16 * 16), 0x00020000);
    const int wvoff = tid * 16;
    auto dma_w = [&](int buf, int kc_) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(&wbuf[buf][wv * 64 + 256 * j]), 16, wvoff,
                                                     (cb * KC + kc_) * (H3_TAP16 * 16) + j * 4096, 0, 0);
    };

    load_t(a.x, areg[0], item, 0, true);
    if (X2) load_t(a.x2, breg, item, 0, true);
    dma_w(0, 0);
    load_t(a.x, areg[1], item, 1, true);                         // KC >= 2
    int par = 0;
    const char* const a0 = reinterpret_cast<const char*>(xin) + (2 * wv * P2_COLS + i) * (CV_PS * 4) + kb * 16;
#pragma unroll 1
    for (; item < nitems; item += gridDim.x) {
        f32x4 mainv[2][4], corrv[2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) { mainv[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; corrv[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        auto chunk = [&](const int kc, f32x4 (&areg)[P2_ROWS]) __attribute__((always_inline)) {
            // (1) this wave's share of the chunk's weights and its A registers have landed -- everything but the P2_ROWS loads of x for the
            //     chunk after this one, which were issued BEHIND this chunk's weight DMA (in-order completion; the order is pinned by the
            //     sched_barrier below and checked in the ISA by tools/isa_scan.py); every wave is past the MFMAs of the chunk before
            //     (the A tile, the other weight buffer and -- after an item's last chunk -- the staging area are free)
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P2_ROWS) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            {
                char* px = reinterpret_cast<char*>(xin) + sc * (CV_PS * 4) + 8 * sq;
#pragma unroll
                for (int u = 0; u < P2_ROWS; ++u) {
                    h4 hi, lo;
                    split4(X2 ? areg[u] + breg[X2 ? u : 0] : areg[u], hi, lo);
                    *reinterpret_cast<h4*>(px + u * (P2_COLS * CV_PS * 4)) = hi;
                    *reinterpret_cast<h4*>(px + u * (P2_COLS * CV_PS * 4) + 128) = lo;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                          // (2) the A tile is complete
            asm volatile("" ::: "memory");
            // requests, in this order: x2 of the chunk after this one; its weights; then (behind them) x of the chunk after that -- of
            // this item or of the workgroup's next one
            const bool more = item + (int)gridDim.x < nitems;
            const bool last = kc + 1 == KC, last2 = kc + 2 >= KC;
            __builtin_amdgcn_sched_barrier(0);
            if (X2) load_t(a.x2, breg, last ? (more ? item + gridDim.x : item) : item, last ? 0 : kc + 1, !last || more);
            __builtin_amdgcn_sched_barrier(0);
            if (!last || more) dma_w(par ^ 1, last ? 0 : kc + 1);
            __builtin_amdgcn_sched_barrier(0);
            load_t(a.x, areg, last2 ? (more ? item + gridDim.x : item) : item, last2 ? kc + 2 - KC : kc + 2, !last2 || more);
            __builtin_amdgcn_sched_barrier(0);
            const char* bp = reinterpret_cast<const char*>(&wbuf[par][0]) + lane * 16;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                h8 ah[2], al[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    ah[mt] = *reinterpret_cast<const h8*>(a0 + mt * (P2_COLS * CV_PS * 4) + 64 * s2);
                    al[mt] = *reinterpret_cast<const h8*>(a0 + mt * (P2_COLS * CV_PS * 4) + 64 * s2 + 128);
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const h8 bh = *reinterpret_cast<const h8*>(bp + 1024 * ((s2 * 4 + nt) * 2));
                    const h8 bl = *reinterpret_cast<const h8*>(bp + 1024 * ((s2 * 4 + nt) * 2 + 1));
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        mainv[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh, mainv[mt][nt], 0, 0, 0);
                        corrv[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bl, corrv[mt][nt], 0, 0, 0);
                        corrv[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], bh, corrv[mt][nt], 0, 0, 0);
                    }
                }
            }
            par ^= 1;
        };
#pragma unroll 1
        for (int kc = 0; kc < KC; kc += 2) {
            chunk(kc, areg[0]);
            chunk(kc + 1, areg[1]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // every wave is done with the A tile: it becomes the staging area
        asm volatile("" ::: "memory");
        // accumulator (reg r, lane (i, kb)) of tile (mt, nt) = pixel (tile row 2 w + mt, column 4 kb + r), column 16 nt + i of the block
        float* stage = xin + wv * (32 * H3_STR);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    stage[(16 * mt + 4 * kb + r) * H3_STR + 16 * nt + i] = fmaf(corrv[mt][nt][r], H3_RSCALE, mainv[mt][nt][r]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            const int t = item / NB, img = t / per_img, trem = t - img * per_img, ty = trem / a.tiles_x;
            const int gy0 = ty * P2_ROWS + 2 * wv, gx0 = (trem - ty * a.tiles_x) * P2_COLS;
            int dy = 0, dx = 0, co0 = 64 * cb;
            if (UP) { const int nco = a.Cout >> 6, q = cb / nco; co0 = 64 * (cb - q * nco); dy = q >> 1; dx = q & 1; }
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)img * a.Hout * a.Wout * a.Cout, 0,
                                                                                (int)((unsigned)a.Hout * (unsigned)a.Wout * (unsigned)pixO), 0x00020000);
            const int l4 = lane >> 4, cq = lane & 15;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int gy = gy0 + (it >> 2), gx = gx0 + 4 * (it & 3) + l4;
                const int oy = UP ? 2 * gy + dy : gy, ox = UP ? 2 * gx + dx : gx;
                const int off = (gy < a.GH && gx < a.GW) ? (oy * a.Wout + ox) * pixO + (co0 + 4 * cq) * 4 : -16;      // outside: dropped
                const f32x4 v = *reinterpret_cast<const f32x4*>(stage + (4 * it + l4) * H3_STR + cq * 4);
                const u32x4v o = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                __builtin_amdgcn_raw_buffer_store_b128(o, ry, off, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // no wave ends with an LDS-DMA in flight
}

// torch weights -> split halves in fragment order, blocks [cb][kc] of 16 KiB (the conv kernel's block format): half j of lane (n, kb) of
// fragment (K step s, N tile nt, part) of block (cb, kc) is part(M[k = 64 kc + 32 s + 8 kb + j][column 64 cb + 16 nt + n]) with
//     down  M[(2 dy + dx) C + ci][co]              = W[co][ci][dy][dx]      (Conv2d weight [2C][C][2][2])
//     up    M[ci][(2 dy + dx) (C / 2) + co]        = W[ci][co][dy][dx]      (ConvTranspose2d weight [C][C/2][2][2])
__global__ __launch_bounds__(256) void k_pix2_pack_w(const float* w, _Float16* wfrag, int C, int up) {
    const int K = up ? C : 4 * C, N = 2 * C, KC = K >> 6;
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;      // one (hi, lo) pair per thread
    if (o >= (long long)K * N) return;
    const int j = o & 7, lane = (o >> 3) & 63, nt = (o >> 9) & 3, s = (o >> 11) & 1;
    const long long blk = o >> 12;                                      // cb * KC + kc
    const int kc = (int)(blk % KC), cb = (int)(blk / KC);
    const int k = 64 * kc + 32 * s + 8 * (lane >> 4) + j, col = 64 * cb + 16 * nt + (lane & 15);
    float v;
    if (up) {
        const int half = C >> 1, q = col / half, co = col - q * half;
        v = w[(((size_t)k * half + co) * 2 + (q >> 1)) * 2 + (q & 1)];
    } else {
        const int q = k / C, ci = k - q * C;
        v = w[(((size_t)col * C + ci) * 2 + (q >> 1)) * 2 + (q & 1)];
    }
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)((v - (float)hi) * H3_SCALE);
    const size_t frag = ((size_t)blk * 2 + s) * 8 + nt * 2;
    wfrag[(frag * 64 + lane) * 8 + j] = hi;
    wfrag[((frag + 1) * 64 + lane) * 8 + j] = lo;
}

hipError_t launch_pix2x2_f16x3(hipStream_t s, const float* x, const float* x2, const float* w, float* y, int n, int C, int H, int W, int up) {
    if (C < 64 || C > 1024 || (C & 63) || (up && (C & 127))) return hipErrorInvalidValue;
    if (!up && ((H | W) & 1)) return hipErrorInvalidValue;
    Pix2Args a;
    a.x = x; a.x2 = x2; a.w = w; a.y = y; a.n = n; a.Hin = H; a.Win = W; a.Cin = C;
    a.Hout = up ? 2 * H : H / 2; a.Wout = up ? 2 * W : W / 2; a.Cout = up ? C / 2 : 2 * C;
    a.GH = up ? H : H / 2; a.GW = up ? W : W / 2;
    a.tiles_x = (a.GW + P2_COLS - 1) / P2_COLS; a.tiles_y = (a.GH + P2_ROWS - 1) / P2_ROWS;
    a.KC = (up ? C : 4 * C) >> 6; a.NB = (2 * C) >> 6;
    if ((long long)a.Hin * a.Win * a.Cin * 4 > 0x7fffffffLL || (long long)a.Hout * a.Wout * a.Cout * 4 > 0x7fffffffLL) return hipErrorInvalidValue;
    const long long items = (long long)n * a.tiles_x * a.tiles_y * a.NB;
    if (items <= 0 || items > 0x7fffffffLL) return hipErrorInvalidValue;
    const int cus = conv_compute_units();
    if (cus <= 0) return hipGetLastError();
    long long grid = 2LL * cus;                                    // persistent workgroups, two per compute unit, a multiple of NB of them
    grid -= grid % a.NB;
    if (grid < a.NB) grid = a.NB;
    if (items < grid) grid = items;                                // items is a multiple of NB as well
    if (up && x2)       hipLaunchKernelGGL((k_pix2x2_h3<true, true>), dim3((unsigned)grid), dim3(CV_THREADS), 0, s, a, (int)items);
    else if (up)        hipLaunchKernelGGL((k_pix2x2_h3<true, false>), dim3((unsigned)grid), dim3(CV_THREADS), 0, s, a, (int)items);
    else if (x2)        hipLaunchKernelGGL((k_pix2x2_h3<false, true>), dim3((unsigned)grid), dim3(CV_THREADS), 0, s, a, (int)items);
    else                hipLaunchKernelGGL((k_pix2x2_h3<false, false>), dim3((unsigned)grid), dim3(CV_THREADS), 0, s, a, (int)items);
    return hipGetLastError();
}

hipError_t launch_pix2_pack_w_f16x3(hipStream_t s, const float* w, float* wfrag, int C, int up) {
    if (C < 64 || C > 1024 || (C & 63) || (up && (C & 127))) return hipErrorInvalidValue;
    const long long pairs = (long long)(up ? C : 4 * C) * 2 * C;
    hipLaunchKernelGGL(k_pix2_pack_w, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, s, w, reinterpret_cast<_Float16*>(wfrag), C, up);
    return hipGetLastError();
}

}  // namespace pnp
