// The 64 -> 64 channel conv3x3 layer of kernels_conv.hip on the HALF-precision matrix cores, float32 results: "f16x3".
//
// gfx950 has no float32 matrix instruction faster than its vector units (v_mfma_f32_32x32x2_f32: 64 FLOP/clk/SIMD), but
// v_mfma_f32_32x32x16_f16 runs at 1024 FLOP/clk/SIMD with EXACT products and float32 accumulation.  Every float32 operand is
// split into two halves,
//
//     x = hi + lo / 2048,   hi = half(x),   lo = half((x - hi) * 2048)          (|x - hi - lo / 2048| <= 2^-22 |x|)
//
// and a product a * b becomes three matrix instructions: hi*hi into one accumulator, hi*lo + lo*hi into a second one that is
// scaled by 1 / 2048 and added at the end (lo*lo, 2^-22 of the product, is dropped).  Half products are exact in float32 (11 x
// 11 significant bits), the accumulation is float32 as in the float32 kernel: the layer's distance from the float64 result
// is that of the float32 kernel (tests/test_gpu_conv.py measures both).  The factor 2048 keeps `lo` a normal half for every
// |x| >= 2^-25; operands beyond +-65504 (the half range) turn into infinities -- loudly, not silently.  Three instructions
// of 16 x the rate: 5.3 x the float32 matrix peak, and the layer becomes a memory-system kernel.
//
// Structure (the float32 kernel's, conv_common.h, where it still fits): persistent 256-thread workgroups, two per compute
// unit, 8 x 16 output pixels x 64 channels per tile, the input tile (halo included) in LDS for the tile's nine taps -- here
// already split: a pixel is [64 hi halves][64 lo halves] + 16 bytes, so an operand fragment (8 consecutive channels of a
// pixel) is one ds_read_b128.  What changes: the float32 kernel streams the weights from L2 into registers, 147 KiB per wave
// and tile -- at this arithmetic rate that would be 85 bytes per clock and compute unit, more than the vector memory path
// delivers.  Here a tap's weights (64 x 64 x [hi, lo] = 16 KiB, in fragment order) are staged through LDS, double-buffered:
// requested two taps ahead into registers, written one tap ahead, ONE barrier per tap.
#include "conv_common.h"

namespace pnp {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr float H3_SCALE = 2048.f, H3_RSCALE = 1.f / 2048.f;
constexpr int H3_TAP16 = 1024;                   // 16-byte units of one tap's weights: [K step 4][N tile 2][hi, lo][lane 64]
template <int DIL> struct GeoH {
    static constexpr int LDS = Geo<DIL>::XIN * 4 + 2 * H3_TAP16 * 16;
    static constexpr int WPS = (CV_MT == 1 && LDS <= 80 * 1024) ? 2 : 1;        // 81 728 bytes at dilation 1: two workgroups fill the 160 KiB
};
static_assert(CV_MT == 1, "the f16x3 kernel is written for one M tile per wave");

__device__ __forceinline__ void split4(const f32x4& v, h4& hi, h4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const _Float16 h = (_Float16)v[e];
        hi[e] = h;
        lo[e] = (_Float16)((v[e] - (float)h) * H3_SCALE);
    }
}

// registers -> LDS tile, split on the way: chunk u of thread tid = 4 consecutive channels of tile pixel p
template <int DIL>
__device__ __forceinline__ void put_input_h3(float* xin, int tid, const Staging<DIL>& st, const f32x4 (&v)[Geo<DIL>::XU]) {
    const int cq = tid & 15;
#pragma unroll
    for (int u = 0; u < Geo<DIL>::XU; ++u)
        if (tid + CV_THREADS * u < Geo<DIL>::HY * Geo<DIL>::HX * 16) {
            h4 hi, lo;
            split4(v[u], hi, lo);
            char* px = reinterpret_cast<char*>(xin + st.loff[u] - 4 * cq);       // the pixel's 272 bytes
            *reinterpret_cast<h4*>(px + 8 * cq) = hi;
            *reinterpret_cast<h4*>(px + 128 + 8 * cq) = lo;
        }
}

template <int DIL>
__global__ __launch_bounds__(CV_THREADS, GeoH<DIL>::WPS) void k_conv3x3_c64_h3(ConvArgs a, int nitems) {
    constexpr int HX = Geo<DIL>::HX;
    __shared__ __attribute__((aligned(16))) float xin[Geo<DIL>::XIN];
    __shared__ __attribute__((aligned(16))) f32x4 wbuf[2][H3_TAP16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kh = lane >> 5;
    const int prow = 2 * wv + (i >> 4), pcol = i & 15;          // this lane's pixel of the wave's 32 (tile coordinates)
    // C = 64 NC channels: an ITEM is (tile, block cb of 64 output channels), its K loop runs over NC chunks of 64 input channels
    // x 9 taps.  item = tile * NC + cb and gridDim.x is a multiple of NC (launch): a workgroup keeps its cb, so its weight
    // stream -- [cb][chunk][tap] blocks of 16 KiB -- is periodic in 9 NC taps, and on an 8-XCD part every XCD works on NC / 8 ..
    // 1 blocks of output channels whose weights (147 KiB x NC) stay in ITS L2 while the input tiles stream.
    const int NC = a.C >> 6, pix = a.C * 4, period = 9 * NC;
    int item = blockIdx.x;
    if (item >= nitems) return;
    const int cb = item % NC;
    const float bias0 = a.bias ? a.bias[64 * cb + i] : 0.f, bias1 = a.bias ? a.bias[64 * cb + i + 32] : 0.f;
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(a.w) + (size_t)cb * period * H3_TAP16 + tid;   // + t * 1024 + 256 j: tap t of the stream

    Staging<DIL> st;
    staging_init<DIL>(a, tid, st, pix);
    f32x4 xpre[Geo<DIL>::XU];
    fetch_input<DIL>(a, tile_pos(a, item / NC), st, xpre, pix, 0);
    f32x4 wreg[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wreg[j] = wsrc[256 * j];
    put_input_h3<DIL>(xin, tid, st, xpre);
#pragma unroll
    for (int j = 0; j < 4; ++j) wbuf[0][tid + 256 * j] = wreg[j];
#pragma unroll
    for (int j = 0; j < 4; ++j) wreg[j] = wsrc[H3_TAP16 + 256 * j];
    int par = 0;                                                 // buffer of the current tap
    int t2 = 2 >= period ? 2 - period : 2;                       // stream position of the weights requested next (two taps ahead)
    // the workgroup that arrived second on its SIMDs starts late, once (kernels_conv.hip): the two stay out of phase
    if (GeoH<DIL>::WPS > 1 && (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1)) __builtin_amdgcn_s_sleep(127);

#pragma unroll 1
    for (; item < nitems; item += gridDim.x) {
        const TilePos q = tile_pos(a, item / NC);
        f32x16 main0, main1, corr0, corr1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { main0[r] = bias0; main1[r] = bias1; corr0[r] = 0.f; corr1[r] = 0.f; }
#pragma unroll 1
        for (int cc = 0; cc < NC; ++cc) {
            // the input tile that follows this one -- the tile's next 64 input channels, or the first 64 of the next item -- is
            // requested now and consumed after this chunk's nine taps
            const bool last = cc + 1 == NC;
            const bool more = !last || item + (int)gridDim.x < nitems;
            if (more) fetch_input<DIL>(a, last ? tile_pos(a, (item + gridDim.x) / NC) : q, st, xpre, pix, last ? 0 : 64 * (cc + 1));
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                // wbuf[par] (and, at tap 0, the input tile) is complete; every wave is done with wbuf[par ^ 1]
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 4; ++j) wbuf[par ^ 1][tid + 256 * j] = wreg[j];        // the next tap's weights
#pragma unroll
                for (int j = 0; j < 4; ++j) wreg[j] = wsrc[t2 * H3_TAP16 + 256 * j];
                t2 = t2 + 1 == period ? 0 : t2 + 1;
                const int ky = tap / 3, kx = tap - 3 * ky;
                const int p = (prow + ky * DIL) * HX + pcol + kx * DIL;                   // input pixel of this tap
                const char* ap = reinterpret_cast<const char*>(xin) + p * (CV_PS * 4) + kh * 16;   // + 32 s: K step s; + 128: the lo halves
                const char* bp = reinterpret_cast<const char*>(&wbuf[par][0]) + lane * 16;        // + 1024 f: fragment f = (2 s + nt) * 2 + part
                h8 ah[2], al[2], bh[2][2], bl[2][2];
#define H3_LOAD(slot, s_)                                                                  \
                ah[slot] = *reinterpret_cast<const h8*>(ap + 32 * (s_));                   \
                al[slot] = *reinterpret_cast<const h8*>(ap + 32 * (s_) + 128);             \
                bh[slot][0] = *reinterpret_cast<const h8*>(bp + 1024 * (4 * (s_) + 0));    \
                bl[slot][0] = *reinterpret_cast<const h8*>(bp + 1024 * (4 * (s_) + 1));    \
                bh[slot][1] = *reinterpret_cast<const h8*>(bp + 1024 * (4 * (s_) + 2));    \
                bl[slot][1] = *reinterpret_cast<const h8*>(bp + 1024 * (4 * (s_) + 3));
                H3_LOAD(0, 0)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int cur = s & 1, nxt = cur ^ 1;
                    if (s + 1 < 4) { H3_LOAD(nxt, s + 1) }
                    __builtin_amdgcn_sched_barrier(0);
                    main0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], bh[cur][0], main0, 0, 0, 0);
                    main1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], bh[cur][1], main1, 0, 0, 0);
                    corr0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], bl[cur][0], corr0, 0, 0, 0);
                    corr1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], bl[cur][1], corr1, 0, 0, 0);
                    corr0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur], bh[cur][0], corr0, 0, 0, 0);
                    corr1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur], bh[cur][1], corr1, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef H3_LOAD
                par ^= 1;
            }
            __syncthreads();                                         // every wave is done with this chunk's input
            if (last) {
                f32x16 acc0[1], acc1[1];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc0[0][r] = fmaf(corr0[r], H3_RSCALE, main0[r]);
                    acc1[0][r] = fmaf(corr1[r], H3_RSCALE, main1[r]);
                }
                store_tile(a, q, xin + wv * (32 * CV_C), wv, lane, acc0, acc1, pix, 64 * cb);
                if (more) __syncthreads();                           // every wave is done with the staging area
            }
            if (more) put_input_h3<DIL>(xin, tid, st, xpre);         // published by the barrier of the next chunk's first tap
        }
    }
}

// torch.nn.Conv2d weight [C out][C in][3][3] -> split halves in fragment order, blocks [cb][chunk cc][tap] of 16 KiB: half j of lane
// (n, kb) of fragment (K step s, N tile nt, part) is part(W[out = 64 cb + 32 nt + n][in = 64 cc + 16 s + 8 kb + j][tap]) --
// v_mfma_f32_32x32x16_f16: lane l supplies B[k = 8 (l >> 5) + j][column l & 31]; the A side reads input channels in the same
// order.  As many bytes as the float32 weights.  Once per model.
__global__ __launch_bounds__(256) void k_conv_pack_w_h3(const float* w_oihw, _Float16* wfrag, int C) {
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;      // one (hi, lo) pair per thread
    if (o >= 9LL * C * C) return;
    const int NC = C >> 6;
    const int j = o & 7, lane = (o >> 3) & 63, nt = (o >> 9) & 1, s = (o >> 10) & 3;
    const long long blk = o >> 12;                                // (cb * NC + cc) * 9 + tap
    const int tap = (int)(blk % 9), cc = (int)((blk / 9) % NC), cb = (int)(blk / (9 * NC));
    const int out = 64 * cb + 32 * nt + (lane & 31), in = 64 * cc + 16 * s + 8 * (lane >> 5) + j;
    const float w = w_oihw[((size_t)out * C + in) * 9 + tap];
    const _Float16 hi = (_Float16)w;
    const _Float16 lo = (_Float16)((w - (float)hi) * H3_SCALE);
    const size_t frag = ((size_t)blk * 4 + s) * 4 + nt * 2;      // the hi fragment; lo follows
    wfrag[(frag * 64 + lane) * 8 + j] = hi;
    wfrag[((frag + 1) * 64 + lane) * 8 + j] = lo;
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
                                int n, int C, int H, int W, int relu, int dilation) {
    if (C < 64 || C > 1024 || (C & 63) || (C != 64 && dilation != 1)) return hipErrorInvalidValue;
    ConvArgs a;
    a.x = x; a.w = w; a.bias = bias; a.skip = skip; a.y = y; a.n = n; a.H = H; a.W = W; a.relu = relu; a.C = C;
    a.tiles_x = (W + CV_TX - 1) / CV_TX; a.tiles_y = (H + CV_TY - 1) / CV_TY;
    const long long items = (long long)n * a.tiles_x * a.tiles_y * (C >> 6);
    if (items <= 0 || items > 0x7fffffffLL) return hipErrorInvalidValue;
    if ((long long)H * W * C * 4 > 0x7fffffffLL) return hipErrorInvalidValue;           // one image must fit a signed 32-bit buffer offset
    const int cus = conv_compute_units();
    if (cus <= 0) return hipGetLastError();
    switch (dilation) {
        case 1: return launch_h3_dil<1>(s, a, items, cus);
        case 2: return launch_h3_dil<2>(s, a, items, cus);
        case 3: return launch_h3_dil<3>(s, a, items, cus);
        case 4: return launch_h3_dil<4>(s, a, items, cus);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_conv_pack_w_f16x3(hipStream_t s, const float* w_oihw, float* wfrag, int C) {
    if (C < 64 || C > 1024 || (C & 63)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_conv_pack_w_h3, dim3((unsigned)(9LL * C * C / 256)), dim3(256), 0, s, w_oihw, reinterpret_cast<_Float16*>(wfrag), C);
    return hipGetLastError();
}

}  // namespace pnp
