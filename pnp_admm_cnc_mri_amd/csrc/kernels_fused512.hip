// Fused gfx950 kernels for 512x512 slices (the shape of BASELINE.json's config 5): the scheme of
// kernels_fused256.hip -- two real slices per complex field, Hermitian k-space blend, mirror-column
// trick, 2 launches per iteration -- with 32-lane cooperative transforms (fft16.h: 512 = 16 points
// x 32 lanes; structure A: t-layout -> k-layout, structure B: k-layout -> t-layout, each in both
// directions).  Layouts: fused_layout.h.
//
//   k5_rows : 8 rows of a slice pair per 256-thread workgroup (a row = 32 consecutive lanes; the
//             radix-2 partner is lane^1: one DPP quad_perm move)
//   k5_cols : 8 column pairs (k2, 512-k2) per workgroup; lane = pair + 8*t_low so that global
//             accesses are 16-byte {P,Q} elements in 128-byte row segments; the radix-2 partner
//             is lane^8 (DPP row_ror:8)
// HBM bytes per slice-iteration: 36 N, as at 256x256.
#include "internal.h"
#include "fused_layout.h"
#include "fused_pointwise.h"
#include <math.h>
#include <stdlib.h>

#ifndef F512_ROWS_W
#define F512_ROWS_W 3       // waves per SIMD of the row / column kernels (experiment knobs; the defaults are the measured best)
#endif
#ifndef F512_COLS_W
#define F512_COLS_W 2
#endif

namespace pnp {

__device__ c32 g_tw512f[512];

struct Fused512 {
    int Bmax = 0, np = 0;
    c32* T = nullptr;
    float4* Yh = nullptr;
    unsigned long long* Mh = nullptr;
    static constexpr int MAXQ = 4;
    hipStream_t side[MAXQ] = {};              // further queues of the chunked schedule (fused512_run); [0] unused
    hipEvent_t ev_fork = nullptr, ev_join[MAXQ] = {};
};

constexpr int NN5 = 512 * 512;

static inline ProxCoef to_coef5(const ProxParams& p) {
    ProxCoef c;
    c.thr = p.thr; c.c1 = p.c1; c.c2 = p.c2; c.c3 = p.c3; c.ib = p.ib;
    return c;
}

__device__ __forceinline__ float dpp_quad_swap1(float v) {      // lane ^ 1
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_row_ror8(float v) {        // lane ^ 8 inside a row of 16 lanes
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x128 /* row_ror:8 */, 0xF, 0xF, true));
}

// ------------------------------------------------------------------------------------------
// table preparation (once per uploaded problem): one block per (tile of 8 columns k2 = 8 m + kl, slice pair); tile 32 = the
// column k2 = 256.  Each slice's y / mask tile and its mirror image (rows -k1, columns -k2) go through LDS, so that global
// memory is read in 64-byte row segments and the table block is written in its own contiguous order (until round 4 a block
// held ONE column and read y with a 4 KiB stride).  Arithmetic = hermitian_entry512 (fused_layout.h).
// ------------------------------------------------------------------------------------------
constexpr int FP5_P = 9;
__global__ __launch_bounds__(256) void k5_prepare(const c32* y, const uint8_t* mask_bank, const int32_t* mask_id,
                                                  float4* Yh, unsigned long long* Mh, int B) {
    __shared__ c32 yd[512 * FP5_P], ym[512 * FP5_P];             // direct tile [row][kl], mirror tile [row][kl] = y[row][-k2]
    __shared__ uint8_t md[512 * FP5_P], mm[512 * FP5_P];
    __shared__ uint8_t nib[512 * 8];                             // [k1][kl]: code of slice a | code of slice b << 2
    const int tid = threadIdx.x, m = blockIdx.x, pair = blockIdx.y;
    const int ncol = (m == F5_TILES - 1) ? 1 : 8;
    c32 ya[16];                                                  // slice a's entries of this thread's 16 output positions
    for (int e = tid; e < 512 * 8; e += 256) nib[e] = 0;
#pragma unroll 1
    for (int sidx = 0; sidx < 2; ++sidx) {
        const int sl = 2 * pair + sidx;
        __syncthreads();
        if (sl < B) {
            const int mid = mask_id ? mask_id[sl] : 0;
            const c32* ys = y + (size_t)sl * NN5;
            const uint8_t* ms = mask_bank + (size_t)mid * NN5;
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const int idx = tid + 256 * i, r = idx >> 3, c = idx & 7;
                if (c < ncol) {
                    const int k2 = 8 * m + c, k2m = (512 - k2) & 511;
                    yd[r * FP5_P + c] = ys[r * 512 + k2];
                    md[r * FP5_P + c] = ms[r * 512 + k2];
                    ym[r * FP5_P + c] = ys[r * 512 + k2m];
                    mm[r * FP5_P + c] = ms[r * 512 + k2m];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int o = tid + 256 * i;                         // storage order inside the tile: [wave 4][q 16][lane 64], lane = kl + 8 (t & 7)
            const int lane = o & 63, q = (o >> 6) & 15, wv = o >> 10;
            const int kl = lane & 7, t = 8 * wv + (lane >> 3);
            const int k1 = (t >> 1) + 16 * q + 256 * (t & 1), r2 = (512 - k1) & 511;       // the k-layout of fft16.h
            c32 yh = mk(0.f, 0.f);
            int code = 0;
            if (sl < B && kl < ncol) {
                const int m1 = md[k1 * FP5_P + kl] != 0, m2 = mm[r2 * FP5_P + kl] != 0;
                const c32 y1 = yd[k1 * FP5_P + kl], y2 = ym[r2 * FP5_P + kl];
                // select, do not multiply: an unsampled y entry (possibly NaN/Inf in user data) must not reach the result
                yh = mk(0.5f * ((m1 ? y1.x : 0.0f) + (m2 ? y2.x : 0.0f)), 0.5f * ((m1 ? y1.y : 0.0f) - (m2 ? y2.y : 0.0f)));
                code = m1 + m2;
            }
            if (kl < ncol) nib[k1 * 8 + kl] |= (uint8_t)(code << (2 * sidx));       // one thread per (k1, kl): no race
            if (sidx == 0) ya[i] = yh;
            else if (kl < ncol) Yh[(size_t)pair * YH5_PAIR + (size_t)m * 4096 + o] = make_float4(ya[i].x, ya[i].y, yh.x, yh.y);
        }
    }
    __syncthreads();
    {
        const int lane = tid & 63, wv = tid >> 6, kl = lane & 7, t = 8 * wv + (lane >> 3);
        if (kl < ncol) {
            unsigned long long v = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) v |= (unsigned long long)nib[((t >> 1) + 16 * q + 256 * (t & 1)) * 8 + kl] << (4 * q);
            Mh[(size_t)pair * MH5_PAIR + (size_t)m * 256 + tid] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// rows
// ------------------------------------------------------------------------------------------

constexpr int RP5 = 520;     // LDS pitch (c32) of a staged row
constexpr int XP5 = 544;     // exchange region per 32-lane group: 16 runs of 34
constexpr int ROWS5_LDS = 8 * XP5 + 544;

// exchanges between the lanes of one 32-lane group through its LDS region (block barriers: all
// 8 groups of the workgroup exchange together)
__device__ __forceinline__ void xchg_t2k(c32 (&a)[16], c32* region, int t) {
    const int k2 = t >> 1, h = t & 1;
#pragma unroll
    for (int k = 0; k < 16; ++k) region[k * 34 + t] = a[k];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = region[k2 * 34 + 2 * i + h];
    __syncthreads();
}
__device__ __forceinline__ void xchg_k2t(c32 (&a)[16], c32* region, int t) {
    const int k2 = t >> 1, h = t & 1;
#pragma unroll
    for (int i = 0; i < 16; ++i) region[k2 * 34 + 2 * i + h] = a[i];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = region[k * 34 + t];
    __syncthreads();
}

// PROX as in kernels_fused256.hip: 0 none, 1 L1 (z and w), 2 CNC, 3 L1 single-state
// 3 waves per SIMD on purpose: with the per-lane twiddle rows the kernel needs <= 128 VGPRs and would run 4, which
// measured 3 % SLOWER on one box (1032 vs 1065 it/s at 256 slices) -- these strided, HBM-bound kernels do not want more waves
template <bool HAS_INV, int PROX, bool HAS_FWD, bool WRITE_X>
__attribute__((amdgpu_waves_per_eu(F512_ROWS_W, F512_ROWS_W)))
__global__ __launch_bounds__(256) void k5_rows(FRowArgs p) {
    __shared__ __attribute__((aligned(16))) c32 lds[ROWS5_LDS];
    const int tid = threadIdx.x, g = tid >> 5, t = tid & 31;
    const int k2 = t >> 1, h = t & 1;
    c32* twl = lds + 8 * XP5;
    twl[17 * (tid >> 4) + (tid & 15)] = g_tw512f[((tid >> 4) * (tid & 15)) & 511];     // [t : 32][k : 16] = W512^(t k), rows of 17: conflict-free
    twl[17 * ((tid >> 4) + 16) + (tid & 15)] = g_tw512f[(((tid >> 4) + 16) * (tid & 15)) & 511];
    const int pair = blockIdx.x >> 6, r0 = (blockIdx.x & 63) * 8;
    const int sa = 2 * pair, sb = sa + 1;
    const bool has_b = sb < p.B;
    c32 a[16];
    c32* Tt = p.T + (size_t)pair * NN5 + (size_t)r0 * 512;       // this block's 8 rows, contiguous
    c32* region = lds + g * XP5;

    if (HAS_INV) {
        const float4* src = reinterpret_cast<const float4*>(Tt);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * i, row = idx >> 8, c2 = idx & 255;
            *reinterpret_cast<float4*>(&lds[row * RP5 + 2 * c2]) = src[idx];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = lds[g * RP5 + phi512(k2 + 16 * q + 256 * h)];
        __syncthreads();
        // structure B, inverse direction: k-layout -> t-layout
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = bfly2(a[q], mk(dpp_quad_swap1(a[q].x), dpp_quad_swap1(a[q].y)), h);
        fft512_b1<true>(a, h);
        xchg_k2t(a, region, t);
        fft512_b2<true>(a, twl + 17 * t);
#pragma unroll
        for (int j = 0; j < 16; ++j) lds[g * RP5 + t + 32 * j] = a[j];
        __syncthreads();
    } else {
        __syncthreads();                                         // twiddle table visible
    }

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i, row = idx >> 7, n4 = (idx & 127) * 4;
        const size_t off = (size_t)(r0 + row) * 512 + n4;
        pointwise4<HAS_INV, PROX, HAS_FWD, WRITE_X>(p, &lds[row * RP5 + n4], (size_t)sa * NN5 + off, (size_t)sb * NN5 + off, has_b);
    }

    if (HAS_FWD) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = lds[g * RP5 + t + 32 * j];
        __syncthreads();
        // structure A, forward direction: t-layout -> k-layout
        fft512_a1<false>(a, twl + 17 * t);
        xchg_t2k(a, region, t);
        fft512_a2<false>(a, h);
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = bfly2(a[q], mk(dpp_quad_swap1(a[q].x), dpp_quad_swap1(a[q].y)), h);
#pragma unroll
        for (int q = 0; q < 16; ++q) lds[g * RP5 + phi512(k2 + 16 * q + 256 * h)] = a[q];
        __syncthreads();
        float4* dst = reinterpret_cast<float4*>(Tt);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * i, row = idx >> 8, c2 = idx & 255;
            dst[idx] = *reinterpret_cast<const float4*>(&lds[row * RP5 + 2 * c2]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// columns
// ------------------------------------------------------------------------------------------
struct F5ColArgs {
    c32* T;
    const float4* Yh;
    const unsigned long long* Mh;
    float c;
};

constexpr int CP5 = 548;     // exchange region (c32) per column group
constexpr int COLS5_LDS = 8 * CP5 + 544;

template <bool INV>
__device__ __forceinline__ void col5_a(c32 (&a)[16], const c32* twl, c32* region, int t) {     // t-layout -> k-layout
    const int h = t & 1;
    fft512_a1<INV>(a, twl + 17 * t);
    xchg_t2k(a, region, t);
    fft512_a2<INV>(a, h);
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = bfly2(a[q], mk(dpp_row_ror8(a[q].x), dpp_row_ror8(a[q].y)), h);
}
template <bool INV>
__device__ __forceinline__ void col5_b(c32 (&a)[16], const c32* twl, c32* region, int t) {     // k-layout -> t-layout
    const int h = t & 1;
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = bfly2(a[q], mk(dpp_row_ror8(a[q].x), dpp_row_ror8(a[q].y)), h);
    fft512_b1<INV>(a, h);
    xchg_k2t(a, region, t);
    fft512_b2<INV>(a, twl + 17 * t);
}

// 2 waves per SIMD on purpose (162 VGPRs would allow 3; measured slower, see k5_rows)
__attribute__((amdgpu_waves_per_eu(F512_COLS_W, F512_COLS_W)))
__global__ __launch_bounds__(256) void k5_cols(F5ColArgs p) {
    __shared__ __attribute__((aligned(16))) c32 lds[COLS5_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int kl = lane & 7, tq = lane >> 3, t = 8 * wv + tq;       // lanes t and t^1 sit 8 apart in one DPP row
    c32* twl = lds + 8 * CP5;
    twl[17 * (tid >> 4) + (tid & 15)] = g_tw512f[((tid >> 4) * (tid & 15)) & 511];     // [t : 32][k : 16] = W512^(t k), rows of 17: conflict-free
    twl[17 * ((tid >> 4) + 16) + (tid & 15)] = g_tw512f[(((tid >> 4) + 16) * (tid & 15)) & 511];
    const int pair = blockIdx.x / 33, m = blockIdx.x % 33;
    // tiles 0..31: column pairs q = 8 m + kl (q >= 1) = physical columns (2q, 2q+1);
    // tile 32: the two self-mirrored columns 0 and 256 (physical 0 and 1), one lane group each.
    const bool self = (m == 32);
    const int k2 = self ? (kl == 0 ? 0 : 256) : 8 * m + kl;
    const bool valid = self ? (kl < 2) : (k2 >= 1);
    c32* Tp = p.T + (size_t)pair * NN5;
    c32 P[16], Q[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        P[j] = mk(0.f, 0.f);
        Q[j] = mk(0.f, 0.f);
        if (valid) {
            const c32* rowp = Tp + (size_t)(t + 32 * j) * 512;
            if (self) {
                P[j] = rowp[kl];
                Q[j] = P[j];
            } else {
                const float4 v = *reinterpret_cast<const float4*>(rowp + 2 * k2);
                P[j] = mk(v.x, v.y);
                Q[j] = mk(v.z, v.w);
            }
        }
    }
    c32* region = lds + kl * CP5;
    __syncthreads();                             // twiddle table visible
    col5_a<false>(P, twl, region, t);            // P[q] = C[k1, k2],   k1 = (t>>1) + 16 q + 256 (t&1)
    col5_a<true>(Q, twl, region, t);             // Q[q] = C[-k1, -k2]
    if (valid) {
        const int m_tab = k2 >> 3, kl_tab = k2 & 7;
        const unsigned long long code = p.Mh[(size_t)pair * MH5_PAIR + ((size_t)(m_tab * 4 + wv) * 64 + kl_tab + 8 * tq)];
        const float4* yhp = p.Yh + (size_t)pair * YH5_PAIR + ((size_t)(m_tab * 4 + wv) * 16) * 64 + kl_tab + 8 * tq;
        const float ch = 0.5f * p.c;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float4 yh = yhp[q * 64];
            const int nibv = (int)((code >> (4 * q)) & 15ull);
            blend_pair(P[q], Q[q], mk(yh.x, yh.y), mk(yh.z, yh.w), nibv & 3, nibv >> 2, p.c, ch);
        }
    }
    col5_b<true>(P, twl, region, t);             // column k2 of the blended field
    col5_b<false>(Q, twl, region, t);            // column 512 - k2
    if (valid) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            c32* rowp = Tp + (size_t)(t + 32 * j) * 512;
            if (self) rowp[kl] = P[j];
            else *reinterpret_cast<float4*>(rowp + 2 * k2) = make_float4(P[j].x, P[j].y, Q[j].x, Q[j].y);
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
Fused512* fused512_create(int Bmax, hipError_t* err) {
    Fused512* f = new Fused512();
    f->Bmax = Bmax;
    f->np = (Bmax + 1) / 2;
    hipError_t e = hipMalloc((void**)&f->T, (size_t)f->np * NN5 * sizeof(c32));
    if (e == hipSuccess) e = hipMalloc((void**)&f->Yh, (size_t)f->np * YH5_PAIR * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc((void**)&f->Mh, (size_t)f->np * MH5_PAIR * sizeof(unsigned long long));
    if (e == hipSuccess) {
        static thread_local c32 h[512];
        for (int m = 0; m < 512; ++m) {
            const double a = -2.0 * M_PI * (double)m / 512.0;
            h[m] = mk((float)cos(a), (float)sin(a));
        }
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_tw512f), h, sizeof(h));
    }
    if (e != hipSuccess) {
        fused512_destroy(f);
        *err = e;
        return nullptr;
    }
    *err = hipSuccess;
    return f;
}

void fused512_destroy(Fused512* f) {
    if (!f) return;
    if (f->T) (void)hipFree(f->T);
    if (f->Yh) (void)hipFree(f->Yh);
    if (f->Mh) (void)hipFree(f->Mh);
    for (int q = 1; q < Fused512::MAXQ; ++q) {
        if (f->side[q]) (void)hipStreamDestroy(f->side[q]);
        if (f->ev_join[q]) (void)hipEventDestroy(f->ev_join[q]);
    }
    if (f->ev_fork) (void)hipEventDestroy(f->ev_fork);
    delete f;
}

hipError_t fused512_prepare(Fused512* f, hipStream_t s, const float2* y, const uint8_t* mask_bank,
                            const int32_t* mask_id, int B) {
    if (B > f->Bmax) return hipErrorInvalidValue;
    const int np = (B + 1) / 2;
    hipLaunchKernelGGL(k5_prepare, dim3(F5_TILES, np), dim3(256), 0, s, reinterpret_cast<const c32*>(y), mask_bank,
                       mask_id, f->Yh, f->Mh, B);
    return hipGetLastError();
}

template <bool HAS_INV, int PROX, bool HAS_FWD, bool WRITE_X>
static hipError_t launch5_rows(hipStream_t s, int np, const FRowArgs& a) {
    hipLaunchKernelGGL((k5_rows<HAS_INV, PROX, HAS_FWD, WRITE_X>), dim3(np * 64), dim3(256), 0, s, a);
    return hipGetLastError();
}

static hipError_t launch5_cols(Fused512* f, hipStream_t s, int pair0, int np, float c) {
    F5ColArgs a;
    a.T = f->T + (size_t)pair0 * NN5;
    a.Yh = f->Yh + (size_t)pair0 * YH5_PAIR;
    a.Mh = f->Mh + (size_t)pair0 * MH5_PAIR;
    a.c = c;
    hipLaunchKernelGGL(k5_cols, dim3(np * 33), dim3(256), 0, s, a);
    return hipGetLastError();
}

static hipError_t run5_chunk(Fused512* f, hipStream_t s, float* z, float* w, float* x, int c0, int Bc, int iters,
                             int prox, float dc_c, const ProxParams& pp) {
    const int np = (Bc + 1) / 2, pair0 = c0 / 2;
    const size_t so = (size_t)c0 * NN5;
    FRowArgs a;
    a.T = f->T + (size_t)pair0 * NN5;
    a.z_in = z + so; a.w_in = w + so; a.z_out = z + so; a.w_out = w + so; a.x_out = x + so; a.B = Bc;
    a.scale = 1.0f / (float)NN5; a.prox = to_coef5(pp); a.u_first = 1;
    hipError_t e = launch5_rows<false, 0, true, false>(s, np, a);
    for (int i = 0; i < iters && e == hipSuccess; ++i) {
        e = launch5_cols(f, s, pair0, np, dc_c);
        if (e != hipSuccess) break;
        const bool last = (i == iters - 1);
        a.u_first = (i == 0);
        if (prox == 2)      e = last ? launch5_rows<true, 2, false, true>(s, np, a) : launch5_rows<true, 2, true, false>(s, np, a);
        else if (prox == 1) e = last ? launch5_rows<true, 1, false, true>(s, np, a) : launch5_rows<true, 1, true, false>(s, np, a);
        else                e = last ? launch5_rows<true, 3, false, true>(s, np, a) : launch5_rows<true, 3, true, false>(s, np, a);
    }
    return e;
}

hipError_t fused512_run(Fused512* f, hipStream_t s, float* z, float* w, float* x, int B, int iters, bool cnc,
                        float dc_c, ProxParams pp, const FusedSchedule& sch) {
    if (iters <= 0) return hipSuccess;
    const int prox = cnc ? 2 : (sch.l1_two_state ? 1 : 3);
    // One queue, sequential launches: measured best at 512x512 (two queues 0.473 ms, mixed launches
    // 0.49 ms against 0.451 ms per iteration at 256 slices -- the 240-VGPR column body would drag
    // the row body down to 2 waves/SIMD in a mixed launch), so the queue / mixed knobs of
    // FusedSchedule apply to the 256x256 path only.
    // chunked round-robin schedule: internal.h, chunk_plan (4 MiB per slice: z, w, T, Yh).  sch.chunk (PNP_FUSED_CHUNK) overrides
    // the chunk size, < 0 = whole batch; sch.chunk_queues overrides the number of queues (experiment builds).
    const ChunkPlan plan = chunk_plan(B, sch, true, false, sch.chunk_queues);
    const int Q = plan.queues, chunk = plan.chunk;
    hipError_t e = hipSuccess;
    if (Q < 2 || B <= chunk) {
        for (int c0 = 0; c0 < B && e == hipSuccess; c0 += chunk)
            e = run5_chunk(f, s, z, w, x, c0, (B - c0 < chunk) ? (B - c0) : chunk, iters, prox, dc_c, pp);
        return e;
    }
    // chunks go round-robin to Q queues: Q chunks in flight, each queue runs all iterations of its chunk before its next one
    if (!f->ev_fork) e = hipEventCreateWithFlags(&f->ev_fork, hipEventDisableTiming);
    for (int q = 1; q < Q && e == hipSuccess; ++q) {
        if (!f->side[q]) {
            e = hipStreamCreateWithFlags(&f->side[q], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_join[q], hipEventDisableTiming);
        }
    }
    if (e != hipSuccess) return e;
    e = hipEventRecord(f->ev_fork, s);
    for (int q = 1; q < Q && e == hipSuccess; ++q) e = hipStreamWaitEvent(f->side[q], f->ev_fork, 0);
    int k = 0;
    for (int c0 = 0; c0 < B && e == hipSuccess; c0 += chunk, ++k) {
        const int q = k % Q;
        e = run5_chunk(f, q ? f->side[q] : s, z, w, x, c0, (B - c0 < chunk) ? (B - c0) : chunk, iters, prox, dc_c, pp);
    }
    for (int q = 1; q < Q && e == hipSuccess; ++q) {
        e = hipEventRecord(f->ev_join[q], f->side[q]);
        if (e == hipSuccess) e = hipStreamWaitEvent(s, f->ev_join[q], 0);
    }
    return e;
}

hipError_t fused512_dc(Fused512* f, hipStream_t s, const float* z, const float* w, float* x, int B, float dc_c) {
    const int np = (B + 1) / 2;
    FRowArgs a;
    a.T = f->T; a.z_in = z; a.w_in = w; a.z_out = nullptr; a.w_out = nullptr; a.x_out = x; a.B = B;
    a.scale = 1.0f / (float)NN5; a.prox = ProxCoef{}; a.u_first = 1;
    hipError_t e = launch5_rows<false, 0, true, false>(s, np, a);
    if (e == hipSuccess) e = launch5_cols(f, s, 0, np, dc_c);
    if (e == hipSuccess) e = launch5_rows<true, 0, false, true>(s, np, a);
    return e;
}

}  // namespace pnp
