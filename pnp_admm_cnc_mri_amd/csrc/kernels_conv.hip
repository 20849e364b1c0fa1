// Optional HIP backend of the denoisers' plain convolution stacks (FFDNet, DnCNN: models/network_ffdnet.py:58-73,
// models/network_dncnn.py:36-67, both built from models/basicblock.py:63-100 `conv(..., mode='CR')`) and of DRUNet's
// 64-channel residual blocks (models/basicblock.py:213-225).  The body layer,
//
//     y = relu?( conv3x3(x, w) + bias + skip ),   64 -> 64 channels, stride 1, zero padding 1, float32,
//
// is an implicit GEMM on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32: exact f32 fma chains at the f32 vector
// peak, 64 FLOP/clk/SIMD -- MI355X has no TF32-like shortcut); the stacks' first (<= 8 -> 64) and last (64 -> <= 4) layers are
// direct kernels at the end of this file.  The north star keeps the CNN forward in PyTorch-ROCm: this is the opt-in
// `Denoiser(backend='hip')`, for the layers where configs 3-5 spend 99.8 % of their time.  DESIGN.md 4.7; every structure
// tried with its time: profiles/conv_variants_r04.txt.
//
//   layout   activations NHWC ([image][row][col][64 channels]): the K direction of the GEMM (input channels of one tap) is
//            contiguous, a pixel is 256 bytes; weights packed once per model into MFMA-fragment order by pnp_conv3x3_c64_pack
//   tile     one 256-thread workgroup = 8 x 16 output pixels x 64 output channels of one image; wave w owns rows 2w, 2w+1
//            (32 pixels = the M of one MFMA) x both halves of the output channels: 2 accumulator tiles = 32 VGPRs.
//            48 KiB of LDS, 172 VGPRs: TWO workgroups per compute unit, one's staging and epilogue run under the other's
//            MFMAs (a 16 x 16 tile with one 512-thread workgroup per unit: 0.70 of the matrix peak)
//   K loop   9 taps x 64 input channels.  The 10 x 18 x 64 input tile (halo included, zeros outside the image) stays in LDS
//            for the whole tile, 272 bytes per pixel (the 16 lanes a b128 read serves per cycle -- consecutive pixels, same
//            channels -- start 4 banks apart; the eight operand groups of a tap are immediate offsets of one address); the
//            weights stream from L2 straight into registers, three operand groups ahead of their use -- no weight staging,
//            no barrier inside a tile's 576 MFMAs per accumulator
//   operands one 16-byte read per lane = four consecutive input channels = the A (or B) operand of FOUR MFMA steps: lane
//            (i, kh) of v_mfma_f32_32x32x2_f32 supplies A[i][k = kh], so lane half kh reads channels 8 g + 4 kh .. + 3 of
//            group g and step e = 0..3 pairs channel 8 g + e with 8 g + 4 + e -- the same on the weight side, so the K order
//            is consistent
//   persist  512 workgroups loop over the tiles; the next tile's input is fetched into registers under this tile's MFMAs
//            (buffer loads whose out-of-image pixels fall outside the descriptor's range and return zeros: no branches)
//   epilogue bias = the accumulators' initial value; the wave's 32 x 64 outputs go through 8 KiB of the idle input tile and
//            leave as eight 16-byte stores per lane (a 256-byte pixel per 16 lanes); skip and ReLU join there
#include "conv_common.h"

namespace pnp {

template <int DIL>
__device__ __forceinline__ void put_input(float* xin, int tid, const Staging<DIL>& st, const f32x4 (&v)[Geo<DIL>::XU]) {
#pragma unroll
    for (int u = 0; u < Geo<DIL>::XU; ++u)
        if (tid + CV_THREADS * u < Geo<DIL>::HY * Geo<DIL>::HX * 16) *reinterpret_cast<f32x4*>(xin + st.loff[u]) = v[u];
}

// Persistent workgroups (two per compute unit): workgroup b works through tiles b, b + gridDim, ...
//   * the INPUT tile (10 x 18 pixels x 64 channels, halo included) lives in LDS; while the matrix cores run tile T, tile
//     T + 1's input is already on its way from memory into registers;
//   * the WEIGHTS never touch LDS: they are packed once per model in MFMA-fragment order (k_conv_pack_w), so that the B
//     operands of one group of four MFMA steps are ONE contiguous 1 KiB wave load per half of the output channels, the same
//     147 KiB for every workgroup (L2-resident), requested CV_BD groups ahead of their use -- an endless periodic stream
//     (group 72 of a tile is group 0 of the next).  With the weight slices staged through LDS a workgroup needed a barrier
//     per tap: 0.78 of the matrix peak, the matrix pipe idle a fifth of the time; without, three barriers per tile.
template <int DIL>
__global__ __launch_bounds__(CV_THREADS, Geo<DIL>::WPS) void k_conv3x3_c64(ConvArgs a, int ntiles) {
    constexpr int HX = Geo<DIL>::HX;
    __shared__ __attribute__((aligned(16))) float xin[Geo<DIL>::XIN];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kh = lane >> 5;
    const int prow = 2 * CV_MT * wv + (i >> 4), pcol = i & 15;  // this lane's pixel of the wave's first M tile (tile coordinates); tile mt: + 2 mt rows
    const float bias0 = a.bias ? a.bias[i] : 0.f, bias1 = a.bias ? a.bias[i + 32] : 0.f;
    const f32x4* wl = reinterpret_cast<const f32x4*>(a.w) + lane;          // + (2 G + nt) * 64: this lane's fragment of group G

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    Staging<DIL> st;
    staging_init<DIL>(a, tid, st);
    f32x4 xpre[Geo<DIL>::XU];
    fetch_input<DIL>(a, tile_pos(a, tile), st, xpre);
    put_input<DIL>(xin, tid, st, xpre);
    // weight fragments of the first CV_BD groups; ring slot = G % 8 (8 groups per tap: the slot of a group is static)
    f32x4 bq0[CV_RING], bq1[CV_RING];
#pragma unroll
    for (int d = 0; d < CV_BD; ++d) { bq0[d] = wl[(2 * d) * 64]; bq1[d] = wl[(2 * d + 1) * 64]; }
    __syncthreads();
    // Two workgroups share a compute unit and run the same program: started together they would load, compute and store
    // together, and the matrix cores would idle through every epilogue.  The one that arrived second on its SIMD (wave slot
    // != 0: HW_REG_HW_ID[3:0]) starts a quarter of a tile late, once; after that the two stay out of phase.  Speed only.
    if (Geo<DIL>::WPS > 1 && (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1)) {
#pragma unroll 1
        for (int k = 0; k < 2; ++k) __builtin_amdgcn_s_sleep(127);
    }

#ifdef CV_PROF
    unsigned long long psum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
#pragma unroll 1
    for (; tile < ntiles; tile += gridDim.x) {
        const TilePos q = tile_pos(a, tile);
        const bool more = tile + (int)gridDim.x < ntiles;
        CV_STAMP(0)
        if (more) fetch_input<DIL>(a, tile_pos(a, tile + gridDim.x), st, xpre); // consumed after this tile's nine taps
        CV_STAMP(1)
        f32x16 acc0[CV_MT], acc1[CV_MT];
#pragma unroll
        for (int mt = 0; mt < CV_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[mt][r] = bias0; acc1[mt][r] = bias1; }     // the bias rides in the accumulators: no add in the epilogue
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int p = (prow + ky * DIL) * HX + pcol + kx * DIL;   // input pixel of this tap
            const float* ap = xin + p * CV_PS + kh * 4;          // + 8 g: the tap's eight operand groups are immediate offsets
            // the weight stream is periodic in 72 groups; (tap * 8 + g + CV_BD) % 72 without a division
            int gpre = tap * 8 + CV_BD;
            // A operands double-buffered in registers: group g + 1's 16-byte read is ISSUED before group g's eight MFMAs and
            // lands under them (left to itself hipcc sinks each read to just before its use)
            f32x4 av[2][CV_MT];
#pragma unroll
            for (int mt = 0; mt < CV_MT; ++mt) av[0][mt] = *reinterpret_cast<const f32x4*>(ap + 2 * mt * HX * CV_PS);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int cur = g & 1, nxt = cur ^ 1;
                if (g + 1 < 8) {
#pragma unroll
                    for (int mt = 0; mt < CV_MT; ++mt) av[nxt][mt] = *reinterpret_cast<const f32x4*>(ap + 2 * mt * HX * CV_PS + 8 * (g + 1));
                }
                {
                    int G = gpre + g;
                    G = G >= CV_GROUPS ? G - CV_GROUPS : G;
                    bq0[(g + CV_BD) & (CV_RING - 1)] = wl[(2 * G) * 64];
                    bq1[(g + CV_BD) & (CV_RING - 1)] = wl[(2 * G + 1) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int mt = 0; mt < CV_MT; ++mt) {
                        acc0[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][mt][e], bq0[g & (CV_RING - 1)][e], acc0[mt], 0, 0, 0);
                        acc1[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][mt][e], bq1[g & (CV_RING - 1)][e], acc1[mt], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        CV_STAMP(2)
        __syncthreads();                                             // every wave is done with this tile's input
        CV_STAMP(3)

        // ---- epilogue (conv_common.h): the wave's 32 x 64 outputs leave through 8 KiB of the (now idle) input tile
        store_tile(a, q, xin + wv * (32 * CV_C), wv, lane, acc0, acc1);
        CV_STAMP(4)
        if (more) {
            __syncthreads();                                         // every wave is done with the staging area
            CV_STAMP(5)
            put_input<DIL>(xin, tid, st, xpre);
            __syncthreads();
            CV_STAMP(6)
        }
    }
#ifdef CV_PROF
    if (tid == 0 && blockIdx.x < 1024) for (int k = 0; k < 8; ++k) g_cvprof[blockIdx.x * 8 + k] = psum[k];
#endif
}

#ifdef CV_PROF
extern "C" int pnp_conv_prof_read(unsigned long long* out /* [1024][8] */) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cvprof), sizeof(unsigned long long) * 1024 * 8);
}
#endif

// torch.nn.Conv2d weight [64 out][64 in][3][3] -> the kernel's fragment order: element e of lane (i, kh) of group G = tap * 8 + g,
// N tile nt is W[out = 32 nt + i][in = 8 g + 4 kh + e][ky][kx], tap = 3 ky + kx  (v_mfma_f32_32x32x2_f32: lane l supplies
// B[k = l >> 5][column l & 31]; the A side reads input channels in the same order).  Once per model.
__global__ __launch_bounds__(256) void k_conv_pack_w(const float* w_oihw, float* wfrag) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= CV_WFRAG) return;
    const int e = o & 3, lane = (o >> 2) & 63, nt = (o >> 8) & 1, G = o >> 9;
    const int tap = G >> 3, g = G & 7, i = lane & 31, kh = lane >> 5;
    const int out = 32 * nt + i, in = 8 * g + 4 * kh + e;
    wfrag[o] = w_oihw[((size_t)out * 64 + in) * 9 + tap];
}

// NCHW <-> NHWC for 64-channel activations (the stacks' first and last layers stay with PyTorch in NCHW): one block per
// (image, 64-pixel run); 64 x 64 tile through LDS, both sides coalesced.
template <bool TO_NHWC>
__global__ __launch_bounds__(256) void k_relayout64(const float* in, float* out, int HW) {
    __shared__ float t[64][65];
    const int img = blockIdx.y, p0 = blockIdx.x * 64, tid = threadIdx.x;
    const float* ib = in + (size_t)img * HW * 64;
    float* ob = out + (size_t)img * HW * 64;
    for (int e = tid; e < 4096; e += 256) {
        const int a_ = e >> 6, b_ = e & 63;                       // input: rows of 64 contiguous elements
        if (TO_NHWC) { const int c = a_, p = p0 + b_; t[c][b_] = p < HW ? ib[(size_t)c * HW + p] : 0.f; }
        else         { const int p = p0 + a_, c = b_; t[b_][a_] = p < HW ? ib[(size_t)p * 64 + c] : 0.f; }
    }
    __syncthreads();
    for (int e = tid; e < 4096; e += 256) {
        const int a_ = e >> 6, b_ = e & 63;
        if (TO_NHWC) { const int p = p0 + a_, c = b_; if (p < HW) ob[(size_t)p * 64 + c] = t[c][a_]; }
        else         { const int c = a_, p = p0 + b_; if (p < HW) ob[(size_t)c * HW + p] = t[c][b_]; }
    }
}

template <int DIL>
static hipError_t launch_conv_dil(hipStream_t s, const ConvArgs& a, long long tiles, int cus) {
    // persistent workgroups (77 KiB of LDS or less: two per compute unit, one otherwise); every workgroup's loop ends: tile < ntiles
    const long long resident = (long long)Geo<DIL>::WPS * cus;
    const unsigned grid = (unsigned)(tiles < resident ? tiles : resident);
    hipLaunchKernelGGL(k_conv3x3_c64<DIL>, dim3(grid), dim3(CV_THREADS), 0, s, a, (int)tiles);
    return hipGetLastError();
}

// compute units of the current device (cached per device): the persistent kernels launch a multiple of it
int conv_compute_units() {
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cus[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}

hipError_t launch_conv3x3_c64(hipStream_t s, const float* x, const float* w, const float* bias, const float* skip, float* y,
                              int n, int H, int W, int relu, int dilation) {
    ConvArgs a;
    a.x = x; a.w = w; a.bias = bias; a.skip = skip; a.y = y; a.n = n; a.H = H; a.W = W; a.relu = relu; a.C = CV_C; a.fmt = 0;
    a.tiles_x = (W + CV_TX - 1) / CV_TX; a.tiles_y = (H + CV_TY - 1) / CV_TY;
    const long long tiles = (long long)n * a.tiles_x * a.tiles_y;
    if (tiles <= 0 || tiles > 0x7fffffffLL) return hipErrorInvalidValue;
    if ((long long)H * W * CV_C * 4 > 0x7fffffffLL) return hipErrorInvalidValue;        // one image must fit a signed 32-bit buffer offset (8 M pixels)
    const int ncu = conv_compute_units();
    if (ncu <= 0) return hipGetLastError();
    switch (dilation) {
        case 1: return launch_conv_dil<1>(s, a, tiles, ncu);
        case 2: return launch_conv_dil<2>(s, a, tiles, ncu);
        case 3: return launch_conv_dil<3>(s, a, tiles, ncu);
        case 4: return launch_conv_dil<4>(s, a, tiles, ncu);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_conv_pack_w(hipStream_t s, const float* w_oihw, float* wfrag) {
    hipLaunchKernelGGL(k_conv_pack_w, dim3(CV_WFRAG / 256), dim3(256), 0, s, w_oihw, wfrag);
    return hipGetLastError();
}

hipError_t launch_relayout64(hipStream_t s, const float* in, float* out, int n, int HW, bool to_nhwc) {
    const dim3 grid((HW + 63) / 64, n);
    if (to_nhwc) hipLaunchKernelGGL(k_relayout64<true>, grid, dim3(256), 0, s, in, out, HW);
    else         hipLaunchKernelGGL(k_relayout64<false>, grid, dim3(256), 0, s, in, out, HW);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------
// First and last layer of the plain stacks (models/network_dncnn.py:52-62, models/network_ffdnet.py:50-56): few input or
// few output channels -- no matrix-core shape, and bound by the 64-channel tensor they write or read (268 MB at 64 images of
// 128 x 128).  Direct convolutions on the vector units, one 256-thread workgroup per 8 x 16 pixel tile, 16 lanes per group of
// 8 consecutive pixels.  With them `Denoiser(backend='hip')` runs DnCNN / FDnCNN / FFDNet without a single MIOpen call.
//   head  x [n][CIN][H][W] (NCHW, CIN <= 8)  ->  y [n][H][W][64] (NHWC), + bias, ReLU: lane cq of a group computes channels
//         4 cq .. 4 cq + 3 of its 8 pixels (acc 32 registers); inputs and weights from LDS (broadcast reads)
//   tail  x [n][H][W][64] (NHWC)  ->  y [n][COUT][H][W] (NCHW, COUT <= 4), + bias: lane cq holds input channels 4 cq .. + 3, the
//         16 partial sums of a pixel meet by four DPP steps inside the lane row
// ------------------------------------------------------------------------------------------
constexpr int CV_HX = Geo<1>::HX, CV_HY = Geo<1>::HY, CV_XU = Geo<1>::XU;      // the direct kernels below: dilation 1
constexpr int HD_MAXC = 8;
struct HeadArgs {
    const float* x; const float* w; const float* bias; float* y;
    int n, cin, H, W, tiles_x, tiles_y, relu;
    // FFDNet's input stage folded into the staging loop (models/network_ffdnet.py:58-68): x is the FULL-resolution image [n][1][src_h][src_w];
    // channels 0..3 of the layer's input are its pixel-unshuffled quarters (channel 2 dy + dx at (y, x) = x[2 y + dy][2 x + dx], replicate-padded
    // to even size = index clamped), channel 4 the noise level sigma[img * sigma_stride]; H = ceil(src_h / 2), W = ceil(src_w / 2)
    int ffdnet, src_h, src_w, sigma_stride;
    const float* sigma;
};
__global__ __launch_bounds__(256) void k_conv3x3_head(HeadArgs a) {
    __shared__ float xin[HD_MAXC * CV_HY * CV_HX];                 // [ci][row 10][col 18]
    __shared__ __attribute__((aligned(16))) float wl[HD_MAXC * 9 * CV_C];   // [ci * 9 + tap][64 out]
    const int tid = threadIdx.x;
    const int per_img = a.tiles_x * a.tiles_y;
    const int img = blockIdx.x / per_img, trem = blockIdx.x - img * per_img, ty = trem / a.tiles_x;
    const int y0 = ty * CV_TY, x0 = (trem - ty * a.tiles_x) * CV_TX;
    const size_t plane = (size_t)a.H * a.W;
    const float* xb = a.ffdnet ? a.x + (size_t)img * a.src_h * a.src_w : a.x + (size_t)img * a.cin * plane;
    const float sig = a.ffdnet ? a.sigma[(size_t)img * a.sigma_stride] : 0.f;
    if (a.ffdnet) {
        // the full-resolution patch, 2 CV_HY rows x 2 CV_HX columns, read row by row (consecutive threads, consecutive pixels) and
        // de-interleaved into the four channel planes on the way into LDS
        for (int e = tid; e < 4 * CV_HY * CV_HX; e += 256) {
            const int pr = e / (2 * CV_HX), pc = e - pr * (2 * CV_HX), r = pr >> 1, c = pc >> 1, ci = 2 * (pr & 1) + (pc & 1);
            const int gy = y0 - 1 + r, gx = x0 - 1 + c;
            const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            const int sy = min(2 * (in ? gy : 0) + (pr & 1), a.src_h - 1), sx = min(2 * (in ? gx : 0) + (pc & 1), a.src_w - 1);
            const float v = xb[(size_t)sy * a.src_w + sx];
            xin[(ci * CV_HY + r) * CV_HX + c] = in ? v : 0.f;
        }
        for (int p = tid; p < CV_HY * CV_HX; p += 256) {           // the convolution zero-pads the noise-level channel too
            const int r = p / CV_HX, c = p - r * CV_HX, gy = y0 - 1 + r, gx = x0 - 1 + c;
            xin[4 * CV_HY * CV_HX + p] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? sig : 0.f;
        }
    } else {
        for (int e = tid; e < a.cin * CV_HY * CV_HX; e += 256) {
            const int ci = e / (CV_HY * CV_HX), p = e - ci * (CV_HY * CV_HX), r = p / CV_HX, c = p - r * CV_HX;
            const int gy = y0 - 1 + r, gx = x0 - 1 + c;
            const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            const float v = xb[(size_t)ci * plane + (size_t)(in ? gy : 0) * a.W + (in ? gx : 0)];
            xin[e] = in ? v : 0.f;
        }
    }
    for (int e = tid; e < a.cin * 9 * CV_C; e += 256) {            // w_oihw [64][cin][3][3] -> [ci * 9 + tap][out]
        const int out = e & 63, k = e >> 6;                        // k = ci * 9 + tap
        wl[e] = a.w[(size_t)out * a.cin * 9 + k];
    }
    __syncthreads();
    const int cq = tid & 15, pg = tid >> 4, row = pg >> 1, col0 = (pg & 1) * 8;
    f32x4 acc[8];
    const f32x4 b4 = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + 4 * cq) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int px = 0; px < 8; ++px) acc[px] = b4;
#pragma unroll 1
    for (int ci = 0; ci < a.cin; ++ci) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            float in[10];
            const float* rp = xin + (ci * CV_HY + row + ky) * CV_HX + col0;
#pragma unroll
            for (int k = 0; k < 10; ++k) in[k] = rp[k];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(wl + (ci * 9 + ky * 3 + kx) * CV_C + 4 * cq);
#pragma unroll
                for (int px = 0; px < 8; ++px) {
                    acc[px][0] = fmaf(in[px + kx], w4[0], acc[px][0]); acc[px][1] = fmaf(in[px + kx], w4[1], acc[px][1]);
                    acc[px][2] = fmaf(in[px + kx], w4[2], acc[px][2]); acc[px][3] = fmaf(in[px + kx], w4[3], acc[px][3]);
                }
            }
        }
    }
    const __amdgpu_buffer_rsrc_t ry = image_rsrc(a.y + (size_t)img * plane * CV_C, a.H, a.W);
    const int gy = y0 + row;
#pragma unroll
    for (int px = 0; px < 8; ++px) {
        const int gx = x0 + col0 + px;
        f32x4 v = acc[px];
        if (a.relu) { v[0] = relu_keep_nan(v[0]); v[1] = relu_keep_nan(v[1]); v[2] = relu_keep_nan(v[2]); v[3] = relu_keep_nan(v[3]); }
        const int off = (gx < a.W) ? (gy * a.W + gx) * (CV_C * 4) + cq * 16 : -16;        // rows below the image: out of range, dropped
        const u32x4v o = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
        __builtin_amdgcn_raw_buffer_store_b128(o, ry, off, 0, 0);
    }
}

constexpr int TL_MAXC = 4;
struct TailArgs {
    const float* x; const float* w; const float* bias; float* y;
    int n, cout, H, W, tiles_x, tiles_y;
};
__device__ __forceinline__ float row16_sum(float v) {             // sum over the 16 lanes of a DPP row, in every lane
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141 /* row_half_mirror */, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140 /* row_mirror */, 0xF, 0xF, true));
    return v;
}
template <int COUT>
__global__ __launch_bounds__(256) void k_conv3x3_tail(TailArgs a) {
    __shared__ __attribute__((aligned(16))) float xin[CV_HY * CV_HX * CV_PS];
    __shared__ __attribute__((aligned(16))) float wl[9 * COUT * CV_C];      // [tap][co][64 in]
    const int tid = threadIdx.x;
    const int per_img = a.tiles_x * a.tiles_y;
    const int img = blockIdx.x / per_img, trem = blockIdx.x - img * per_img, ty = trem / a.tiles_x;
    const int y0 = ty * CV_TY, x0 = (trem - ty * a.tiles_x) * CV_TX;
    const size_t plane = (size_t)a.H * a.W;
    {
        const __amdgpu_buffer_rsrc_t rs = image_rsrc(a.x + (size_t)img * plane * CV_C, a.H, a.W);
        const int origin = ((y0 - 1) * a.W + (x0 - 1)) * (CV_C * 4);
#pragma unroll
        for (int u = 0; u < CV_XU; ++u) {
            const int idx = tid + 256 * u, p = idx >> 4, cq = idx & 15, r = p / CV_HX, c = p - r * CV_HX;
            const int gx = x0 - 1 + c;
            const bool any = idx < CV_HY * CV_HX * 16;
            const int off = (any && gx >= 0 && gx < a.W) ? origin + (r * a.W + c) * (CV_C * 4) + cq * 16 : -16;
            const u32x4v w = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
            if (any) *reinterpret_cast<f32x4*>(xin + p * CV_PS + cq * 4) = f32x4{__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w)};
        }
    }
    for (int e = tid; e < 9 * COUT * CV_C; e += 256) {             // w_oihw [cout][64][3][3] -> [tap][co][in]
        const int in = e & 63, co = (e >> 6) % COUT, tap = e / (64 * COUT);
        wl[e] = a.w[((size_t)co * 64 + in) * 9 + tap];
    }
    __syncthreads();
    const int cq = tid & 15, pg = tid >> 4, row = pg >> 1, col0 = (pg & 1) * 8;
    float acc[8][COUT];
#pragma unroll
    for (int px = 0; px < 8; ++px)
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[px][co] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        f32x4 in[10];
        const float* rp = xin + ((row + ky) * CV_HX + col0) * CV_PS + cq * 4;
#pragma unroll
        for (int k = 0; k < 10; ++k) in[k] = *reinterpret_cast<const f32x4*>(rp + k * CV_PS);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(wl + ((ky * 3 + kx) * COUT + co) * CV_C + cq * 4);
#pragma unroll
                for (int px = 0; px < 8; ++px) {
                    const f32x4 v = in[px + kx];
                    acc[px][co] = fmaf(v[3], w4[3], fmaf(v[2], w4[2], fmaf(v[1], w4[1], fmaf(v[0], w4[0], acc[px][co]))));
                }
            }
        }
    }
    // the 16 lanes of a group hold partial sums over their four input channels: total them, lane px (< 8) keeps pixel px
    const int gy = y0 + row, gx = x0 + col0 + cq;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        float mine = 0.f;
#pragma unroll
        for (int px = 0; px < 8; ++px) {
            const float s = row16_sum(acc[px][co]);
            mine = (cq == px) ? s : mine;
        }
        if (cq < 8 && gy < a.H && gx < a.W)
            a.y[((size_t)img * COUT + co) * plane + (size_t)gy * a.W + gx] = mine + (a.bias ? a.bias[co] : 0.f);
    }
}

hipError_t launch_conv3x3_head(hipStream_t s, const float* x_nchw, const float* w_oihw, const float* bias, float* y_nhwc,
                               int n, int cin, int H, int W, int relu) {
    if (cin < 1 || cin > HD_MAXC || (long long)H * W * CV_C * 4 > 0x7fffffffLL) return hipErrorInvalidValue;
    HeadArgs a;
    a.x = x_nchw; a.w = w_oihw; a.bias = bias; a.y = y_nhwc; a.n = n; a.cin = cin; a.H = H; a.W = W; a.relu = relu;
    a.ffdnet = 0; a.src_h = a.src_w = a.sigma_stride = 0; a.sigma = nullptr;
    a.tiles_x = (W + CV_TX - 1) / CV_TX; a.tiles_y = (H + CV_TY - 1) / CV_TY;
    const long long tiles = (long long)n * a.tiles_x * a.tiles_y;
    if (tiles <= 0 || tiles > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_conv3x3_head, dim3((unsigned)tiles), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_ffdnet_head(hipStream_t s, const float* x_full, const float* sigma, int sigma_per_image, const float* w_oihw, const float* bias,
                              float* y_nhwc, int n, int h, int w, int relu) {
    const int H = (h + 1) / 2, W = (w + 1) / 2;
    if (h < 1 || w < 1 || (long long)H * W * CV_C * 4 > 0x7fffffffLL) return hipErrorInvalidValue;
    HeadArgs a;
    a.x = x_full; a.w = w_oihw; a.bias = bias; a.y = y_nhwc; a.n = n; a.cin = 5; a.H = H; a.W = W; a.relu = relu;
    a.ffdnet = 1; a.src_h = h; a.src_w = w; a.sigma = sigma; a.sigma_stride = sigma_per_image ? 1 : 0;
    a.tiles_x = (W + CV_TX - 1) / CV_TX; a.tiles_y = (H + CV_TY - 1) / CV_TY;
    const long long tiles = (long long)n * a.tiles_x * a.tiles_y;
    if (tiles <= 0 || tiles > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_conv3x3_head, dim3((unsigned)tiles), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_conv3x3_tail(hipStream_t s, const float* x_nhwc, const float* w_oihw, const float* bias, float* y_nchw,
                               int n, int cout, int H, int W) {
    if (cout < 1 || cout > TL_MAXC || (long long)H * W * CV_C * 4 > 0x7fffffffLL) return hipErrorInvalidValue;
    TailArgs a;
    a.x = x_nhwc; a.w = w_oihw; a.bias = bias; a.y = y_nchw; a.n = n; a.cout = cout; a.H = H; a.W = W;
    a.tiles_x = (W + CV_TX - 1) / CV_TX; a.tiles_y = (H + CV_TY - 1) / CV_TY;
    const long long tiles = (long long)n * a.tiles_x * a.tiles_y;
    if (tiles <= 0 || tiles > 0x7fffffffLL) return hipErrorInvalidValue;
    const dim3 grid((unsigned)tiles), block(256);
    switch (cout) {
        case 1: hipLaunchKernelGGL(k_conv3x3_tail<1>, grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL(k_conv3x3_tail<2>, grid, block, 0, s, a); break;
        case 3: hipLaunchKernelGGL(k_conv3x3_tail<3>, grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL(k_conv3x3_tail<4>, grid, block, 0, s, a); break;
    }
    return hipGetLastError();
}

}  // namespace pnp
