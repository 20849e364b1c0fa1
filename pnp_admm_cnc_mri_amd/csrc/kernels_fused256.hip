// Fused gfx950 kernels for 256x256 slices: one ADMM iteration = 2 launches per part of the batch
// (sequential schedule: k_frows + k_fcols; default schedule: two k_fmixed launches per queue, each
// holding the row workgroups of one half of the part and the column workgroups of the other).
//
//   k_frows : per 16 rows of a slice PAIR (a = 2p, b = 2p+1, carried as c = v_a + i v_b):
//             [inverse row FFT of T -> x = |Re|,|Im| / 65536 -> L1/CNC z-update -> dual update]
//             -> v = z - w -> forward row FFT -> T          (reference lines S4:119-132)
//   k_fcols : per 16 column pairs (k2, 256-k2) of a slice pair:
//             forward column FFT of column k2, inverse-direction FFT of column 256-k2 (so that
//             lane/register (t, j) holds C[k] and C[-k] together -- no exchange for the
//             two-slice unpack), Hermitian data-consistency blend against Yh/Mh, and the
//             inverse column transforms, in place in T.   (S4:120-123)
//
// HBM bytes per slice-iteration: z,w read+write 16 N + T write/read twice 16 N (8 N per slice
// as two slices share a complex field) + Yh 4 N + Mh ~0 = 36 N, against 57 N for the plain c2c
// three-kernel formulation.  Layouts: fused_layout.h; cores: fft16.h.
//
// FFT-256 = 16 lanes x 16 points (radix-16 in registers, one 16x16 transpose through LDS).
//   rows   : lanes of a 16-group are consecutive (one row each), global access is 16 B/lane
//            through an LDS staging tile;
//   columns: a group's lanes sit 16 apart (lane = pair_local + 16*t_quad, 4 waves) so every global
//            access instruction covers 256-B row segments: 16 neighbouring {C[r][q], C[r][256-q]}
//            elements of 16 bytes (mirrored columns are stored side by side, see phi()).
// The W256 twiddle table is staged in LDS and read where used; keeping it in 32 VGPRs cost a wave
// per SIMD.  Everything is compiled with -ffp-contract=off and explicit fmaf (fft16.h), so all
// kernel variants and schedules round identically.
#include "internal.h"
#include "fused_layout.h"
#include "fused_pointwise.h"
#include <math.h>
#include <stdlib.h>

namespace pnp {

__device__ c32 g_twf[256];
__device__ c64 g_twd[256];
template <typename R> __device__ __forceinline__ const cxT<R>* tw_table();
template <> __device__ __forceinline__ const c32* tw_table<float>() { return g_twf; }
template <> __device__ __forceinline__ const c64* tw_table<double>() { return g_twd; }

struct Fused256 {
    int Bmax = 0, np = 0;
    c32* T = nullptr;
    float4* Yh = nullptr;
    unsigned long long* Mh = nullptr;
    // extra queues: parts of the batch run their whole K-iteration chains concurrently, so the
    // bandwidth-bound row kernel of one part fills the memory-idle phases of another part's
    // column kernel (slices are independent; results do not depend on the split)
    static constexpr int MAXQ = 4;
    hipStream_t side[MAXQ - 1] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr;
    hipEvent_t ev_join[MAXQ - 1] = {nullptr, nullptr, nullptr};
};

static inline ProxCoef to_coef(const ProxParams& p) {
    ProxCoef c;
    c.thr = p.thr; c.c1 = p.c1; c.c2 = p.c2; c.c3 = p.c3; c.ib = p.ib;
    return c;
}

// ------------------------------------------------------------------------------------------
// table preparation (once per uploaded problem)
// ------------------------------------------------------------------------------------------
// One block per (tile of 16 columns k2, slice pair); tile 8 = the column k2 = 128.  Each slice's y / mask tile and its mirror
// image (rows -k1, columns -k2) go through LDS, so that global memory is read in 128-byte row segments and the table is
// written in its own contiguous order (until round 3 a block held ONE column and read y with a 2 KiB stride: 2.5 GB fetched
// for 0.4 GB of input at 512 slices).  Arithmetic = hermitian_entry (fused_layout.h).
constexpr int FP_P = 17;
__global__ __launch_bounds__(256) void k_fprepare(const c32* y, const uint8_t* mask_bank, const int32_t* mask_id,
                                                  float4* Yh, unsigned long long* Mh, int B) {
    __shared__ c32 yd[256 * FP_P], ym[256 * FP_P];
    __shared__ uint8_t md[256 * FP_P], mm[256 * FP_P];
    __shared__ uint8_t nib[256 * 16];                            // [k1][c]: code of slice a | code of slice b << 2
    const int tid = threadIdx.x, m = blockIdx.x, pair = blockIdx.y;
    const int ncol = (m == 8) ? 1 : 16;
    c32 ya[16];                                                  // slice a's entries of this thread's 16 output positions
    for (int e = tid; e < 256 * 16; e += 256) nib[e] = 0;
#pragma unroll 1
    for (int sidx = 0; sidx < 2; ++sidx) {
        const int sl = 2 * pair + sidx;
        __syncthreads();
        if (sl < B) {
            const int mid = mask_id ? mask_id[sl] : 0;
            const c32* ys = y + (size_t)sl * 65536;
            const uint8_t* ms = mask_bank + (size_t)mid * 65536;
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const int idx = tid + 256 * i, r = idx >> 4, c = idx & 15;
                if (c < ncol) {
                    const int k2 = 16 * m + c, k2m = (256 - k2) & 255;
                    yd[r * FP_P + c] = ys[r * 256 + k2];
                    md[r * FP_P + c] = ms[r * 256 + k2];
                    ym[r * FP_P + c] = ys[r * 256 + k2m];
                    mm[r * FP_P + c] = ms[r * 256 + k2m];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int o = tid + 256 * i;                         // storage order inside the tile: [wave 4][j 16][kl 16][tq 4]
            const int tq = o & 3, kl = (o >> 2) & 15, j = (o >> 6) & 15, wv = o >> 10;
            const int k1 = 4 * wv + tq + 16 * j, r2 = (256 - k1) & 255;
            c32 yh = mk(0.f, 0.f);
            int code = 0;
            if (sl < B && kl < ncol) {
                const int m1 = md[k1 * FP_P + kl] != 0, m2 = mm[r2 * FP_P + kl] != 0;
                const c32 y1 = yd[k1 * FP_P + kl], y2 = ym[r2 * FP_P + kl];
                // select, do not multiply: an unsampled y entry (possibly NaN/Inf in user data) must not reach the result
                yh = mk(0.5f * ((m1 ? y1.x : 0.0f) + (m2 ? y2.x : 0.0f)), 0.5f * ((m1 ? y1.y : 0.0f) - (m2 ? y2.y : 0.0f)));
                code = m1 + m2;
            }
            if (kl < ncol) nib[k1 * 16 + kl] |= (uint8_t)(code << (2 * sidx));      // one thread per (k1, kl): no race
            if (sidx == 0) ya[i] = yh;
            else if (kl < ncol) Yh[(size_t)pair * YH_PAIR + (size_t)m * 4096 + o] = make_float4(ya[i].x, ya[i].y, yh.x, yh.y);
        }
    }
    __syncthreads();
    {
        const int tq = tid & 3, kl = (tid >> 2) & 15, wv = tid >> 6, t = 4 * wv + tq;
        if (kl < ncol) {
            unsigned long long v = 0;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) v |= (unsigned long long)nib[(t + 16 * jj) * 16 + kl] << (4 * jj);
            Mh[(size_t)pair * MH_PAIR + (size_t)m * 256 + tid] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// rows
// ------------------------------------------------------------------------------------------

// physical column of k-space column k inside a row of T: mirrored columns sit side by side so the
// column kernel moves {C[r][q], C[r][256-q]} as ONE aligned 16-byte access:
//   [0]=col 0, [1]=col 128, [2q]=col q, [2q+1]=col 256-q   (q = 1..127)
__device__ __forceinline__ int phi(int k) { return k < 128 ? 2 * k : (k == 128 ? 1 : 513 - 2 * k); }

template <bool INV, typename R>
__device__ __forceinline__ void fft256_head_lds(cxT<R> (&a)[16], const cxT<R>* twl, int t) {
    dft16<INV>(a);
#pragma unroll
    for (int k = 1; k < 16; ++k) a[k] = tmul<INV>(a[k], twl[17 * t + k]);     // one address register + immediate offsets
}

constexpr int RP = 272;    // staging pitch (c32) of a row in LDS: 272 % 32 == 16 -> two rows per
                           // 32-lane ds_read_b64 group land on disjoint bank halves
constexpr int XP = 289;    // exchange region per 16-lane group, runs of 17 (289 % 32 == 1)

// 16x16 transpose between the lanes of a group; region = this group's XP-sized LDS area
template <bool INV, typename R>
__device__ __forceinline__ void row_fft256(cxT<R> (&a)[16], const cxT<R>* twl, cxT<R>* region, int t) {
    fft256_head_lds<INV>(a, twl, t);
#pragma unroll
    for (int k = 0; k < 16; ++k) region[k * 17 + t] = a[k];
    __syncthreads();
#pragma unroll
    for (int n = 0; n < 16; ++n) a[n] = region[t * 17 + n];
    __syncthreads();
    fft256_tail<INV>(a);
}

// PROX: see fused_pointwise.h
constexpr int ROWS_LDS = 16 * XP + 272;      // c32 elements of LDS the row body needs (exchange regions + 16 x 17 twiddle rows)

template <typename R, bool HAS_INV, int PROX, bool HAS_FWD, bool WRITE_X>
__device__ __forceinline__ void frows_body(const FRowArgsT<R>& p, const int bid, cxT<R>* lds) {
    using C = cxT<R>;
    struct alignas(16) U16 { char b[16]; };              // one 16-byte global / LDS access
    constexpr int EPU = 16 / (int)sizeof(C);             // elements per access: 2 (float) or 1 (double)
    constexpr int UPR = 256 / EPU;                       // accesses per row
    constexpr int NU = 16 * UPR / 256;                   // accesses per thread for the block's 16 rows
    const int tid = threadIdx.x, g = tid >> 4, t = tid & 15;
    C* twl = lds + 16 * XP;                    // W256 table (read where used, not held in VGPRs)
    twl[17 * (tid >> 4) + (tid & 15)] = tw_table<R>()[((tid >> 4) * (tid & 15)) & 255];   // [t][k] = W256^(t k), rows of 17: conflict-free
    const int pair = bid >> 4, r0 = (bid & 15) * 16;
    const int sa = 2 * pair, sb = sa + 1;
    const bool has_b = sb < p.B;
    C a[16];
    C* Tt = p.T + (size_t)pair * 65536 + (size_t)r0 * 256;       // this block's 16 rows, contiguous

    if (HAS_INV) {
        const U16* src = reinterpret_cast<const U16*>(Tt);
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int idx = tid + 256 * i, row = idx / UPR, c2 = idx % UPR;
            *reinterpret_cast<U16*>(&lds[row * RP + EPU * c2]) = src[idx];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = lds[g * RP + phi(t + 16 * j)];
        __syncthreads();
        row_fft256<true>(a, twl, lds + g * XP, t);
#pragma unroll
        for (int j = 0; j < 16; ++j) lds[g * RP + t + 16 * j] = a[j];
        __syncthreads();
    }

    // pointwise phase, 4 consecutive pixels per lane (16-B global accesses)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i, row = idx >> 6, n4 = (idx & 63) * 4;
        const size_t off = (size_t)(r0 + row) * 256 + n4;
        pointwise4<HAS_INV, PROX, HAS_FWD, WRITE_X>(p, &lds[row * RP + n4], (size_t)sa * 65536 + off, (size_t)sb * 65536 + off, has_b);
    }

    if (HAS_FWD) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = lds[g * RP + t + 16 * j];
        __syncthreads();
        row_fft256<false>(a, twl, lds + g * XP, t);
#pragma unroll
        for (int j = 0; j < 16; ++j) lds[g * RP + phi(t + 16 * j)] = a[j];
        __syncthreads();
        U16* dst = reinterpret_cast<U16*>(Tt);
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int idx = tid + 256 * i, row = idx / UPR, c2 = idx % UPR;
            dst[idx] = *reinterpret_cast<const U16*>(&lds[row * RP + EPU * c2]);
        }
    }
}

template <typename R, bool HAS_INV, int PROX, bool HAS_FWD, bool WRITE_X>
__global__ __launch_bounds__(256) void k_frows(FRowArgsT<R> p) {
    __shared__ __attribute__((aligned(16))) cxT<R> lds[ROWS_LDS];
    frows_body<R, HAS_INV, PROX, HAS_FWD, WRITE_X>(p, blockIdx.x, lds);
}

// ------------------------------------------------------------------------------------------
// columns
// ------------------------------------------------------------------------------------------
struct FColArgs {
    c32* T;
    const float4* Yh;
    const unsigned long long* Mh;
    float c;
};

constexpr int CP = 257;     // exchange region (c32) per column group: 514 dwords % 64 == 2

template <bool INV, typename R>
__device__ __forceinline__ void col_exchange(cxT<R> (&a)[16], cxT<R>* region, int t) {
#pragma unroll
    for (int k = 0; k < 16; ++k) region[k * 16 + t] = a[k];
    __syncthreads();
#pragma unroll
    for (int n = 0; n < 16; ++n) a[n] = region[t * 16 + n];
    __syncthreads();
}

constexpr int COLS_LDS = 16 * CP + 272;

__device__ __forceinline__ void fcols_body(const FColArgs& p, const int bid, c32* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int kl = lane & 15, t = 4 * wv + (lane >> 4);
    const int pair = bid / 9, m = bid % 9;
    // tiles 0..7: column pairs q = 16 m + kl (q >= 1) = physical columns (2q, 2q+1);
    // tile 8: the two self-mirrored columns 0 and 128 (physical 0 and 1), one lane group each.
    const bool self = (m == 8);
    const int k2 = self ? (kl == 0 ? 0 : 128) : 16 * m + kl;
    const bool valid = self ? (kl < 2) : (k2 >= 1);
    c32* Tp = p.T + (size_t)pair * 65536;
    // W256 table in LDS: twiddles are fetched where they are used (ds_read_b64, 4 distinct
    // addresses per wave) instead of occupying 32 VGPRs for the whole kernel
    c32* twl = lds + 16 * CP;
    twl[17 * (tid >> 4) + (tid & 15)] = g_twf[((tid >> 4) * (tid & 15)) & 255];            // [t][k] = W256^(t k), rows of 17: conflict-free
    c32 P[16], Q[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        P[j] = mk(0.f, 0.f);
        Q[j] = mk(0.f, 0.f);
        if (valid) {
            const c32* rowp = Tp + (t + 16 * j) * 256;
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
    c32* region = lds + kl * CP;
    __syncthreads();                             // twiddle table visible
    fft256_head_lds<false>(P, twl, t);
    col_exchange<false>(P, region, t);
    fft256_tail<false>(P);                       // P[j] = C[k1 = t + 16 j, k2]
    fft256_head_lds<true>(Q, twl, t);
    col_exchange<true>(Q, region, t);
    fft256_tail<true>(Q);                        // Q[j] = C[-k1, -k2]
    if (valid) {
        const unsigned long long code = p.Mh[mh_index(pair, k2, t)];
        const float ch = 0.5f * p.c;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float4 yh = p.Yh[yh_index(pair, k2, j, t)];
            const int nibv = (int)((code >> (4 * j)) & 15ull);
            blend_pair(P[j], Q[j], mk(yh.x, yh.y), mk(yh.z, yh.w), nibv & 3, nibv >> 2, p.c, ch);
        }
    }
    fft256_head_lds<true>(P, twl, t);
    col_exchange<true>(P, region, t);
    fft256_tail<true>(P);                        // column k2 of the blended field
    fft256_head_lds<false>(Q, twl, t);
    col_exchange<false>(Q, region, t);
    fft256_tail<false>(Q);                       // column 256 - k2
    if (valid) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            c32* rowp = Tp + (t + 16 * j) * 256;
            if (self) rowp[kl] = P[j];
            else *reinterpret_cast<float4*>(rowp + 2 * k2) = make_float4(P[j].x, P[j].y, Q[j].x, Q[j].y);
        }
    }
}

__global__ __launch_bounds__(256) void k_fcols(FColArgs p) {
    __shared__ __attribute__((aligned(16))) c32 lds[COLS_LDS];
    fcols_body(p, blockIdx.x, lds);
}

// ------------------------------------------------------------------------------------------
// mixed launch: ONE grid that holds the row workgroups of one half of the batch and the column
// workgroups of the other half, dealt 16 : 9 so that every CU hosts both kinds at once.  The row
// body streams at the HBM rate while the column body has memory-idle phases (operand latency,
// FFT arithmetic, barriers); co-resident, the two fill each other's gaps -- deterministically,
// instead of hoping two HIP queues interleave.  nR = 16 * pairs(rows half), nC = 9 * pairs(cols half).
// ------------------------------------------------------------------------------------------
template <bool HAS_INV, int PROX, bool HAS_FWD, bool WRITE_X>
__global__ __launch_bounds__(256) void k_fmixed(FRowArgs pr, FColArgs pc, int nR, int nC) {
    __shared__ __attribute__((aligned(16))) c32 lds[ROWS_LDS > COLS_LDS ? ROWS_LDS : COLS_LDS];
    const int b = blockIdx.x, g = b / 25, r = b % 25;
    // groups of 25 consecutive blocks = 16 row + 9 column workgroups while both kinds last
    const int full = (nR / 16 < nC / 9) ? nR / 16 : nC / 9;
    int rid = -1, cid = -1;
    if (g < full) {
        if (r < 16) rid = g * 16 + r; else cid = g * 9 + (r - 16);
    } else {
        const int rest = b - full * 25;                  // leftovers of the longer kind
        if (nR > full * 16) rid = full * 16 + rest; else cid = full * 9 + rest;
    }
    if (rid >= 0) { if (rid < nR) frows_body<float, HAS_INV, PROX, HAS_FWD, WRITE_X>(pr, rid, lds); }
    else if (cid < nC) fcols_body(pc, cid, lds);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
Fused256* fused256_create(int Bmax, hipError_t* err) {
    Fused256* f = new Fused256();
    f->Bmax = Bmax;
    f->np = (Bmax + 1) / 2;
    hipError_t e = hipMalloc((void**)&f->T, (size_t)f->np * 65536 * sizeof(c32));
    if (e == hipSuccess) e = hipMalloc((void**)&f->Yh, (size_t)f->np * YH_PAIR * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc((void**)&f->Mh, (size_t)f->np * MH_PAIR * sizeof(unsigned long long));
    if (e == hipSuccess) {
        static thread_local c32 h[256];
        for (int m = 0; m < 256; ++m) {
            const double a = -2.0 * M_PI * (double)m / 256.0;
            h[m] = mk((float)cos(a), (float)sin(a));
        }
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_twf), h, sizeof(h));
    }
    if (e != hipSuccess) {
        fused256_destroy(f);
        *err = e;
        return nullptr;
    }
    *err = hipSuccess;
    return f;
}

void fused256_destroy(Fused256* f) {
    if (!f) return;
    if (f->T) (void)hipFree(f->T);
    if (f->Yh) (void)hipFree(f->Yh);
    if (f->Mh) (void)hipFree(f->Mh);
    for (int q = 0; q < Fused256::MAXQ - 1; ++q) {
        if (f->side[q]) (void)hipStreamDestroy(f->side[q]);
        if (f->ev_join[q]) (void)hipEventDestroy(f->ev_join[q]);
    }
    if (f->ev_fork) (void)hipEventDestroy(f->ev_fork);
    delete f;
}

hipError_t fused256_prepare(Fused256* f, hipStream_t s, const float2* y, const uint8_t* mask_bank,
                            const int32_t* mask_id, int B) {
    if (B > f->Bmax) return hipErrorInvalidValue;
    const int np = (B + 1) / 2;
    hipLaunchKernelGGL(k_fprepare, dim3(F_TILES, np), dim3(256), 0, s, reinterpret_cast<const c32*>(y), mask_bank,
                       mask_id, f->Yh, f->Mh, B);
    return hipGetLastError();
}

template <bool HAS_INV, int PROX, bool HAS_FWD, bool WRITE_X>
static hipError_t launch_frows(hipStream_t s, int np, const FRowArgs& a) {
    hipLaunchKernelGGL((k_frows<float, HAS_INV, PROX, HAS_FWD, WRITE_X>), dim3(np * 16), dim3(256), 0, s, a);
    return hipGetLastError();
}

static FColArgs col_args(Fused256* f, int pair0, float c) {
    FColArgs a;
    a.T = f->T + (size_t)pair0 * 65536;
    a.Yh = f->Yh + (size_t)pair0 * YH_PAIR;
    a.Mh = f->Mh + (size_t)pair0 * MH_PAIR;
    a.c = c;
    return a;
}

static hipError_t launch_fcols(Fused256* f, hipStream_t s, int pair0, int np, float c) {
    const FColArgs a = col_args(f, pair0, c);
    hipLaunchKernelGGL(k_fcols, dim3(np * 9), dim3(256), 0, s, a);
    return hipGetLastError();
}

// K iterations on slices [c0, c0+Bc) enqueued on stream s
static hipError_t run_chunk(Fused256* f, hipStream_t s, float* z, float* w, float* x, int c0, int Bc, int iters,
                            int prox, float dc_c, const ProxParams& pp) {
    const int np = (Bc + 1) / 2, pair0 = c0 / 2;
    const size_t so = (size_t)c0 * 65536;
    FRowArgs a;
    a.T = f->T + (size_t)pair0 * 65536;
    a.z_in = z + so; a.w_in = w + so; a.z_out = z + so; a.w_out = w + so; a.x_out = x + so; a.B = Bc;
    a.scale = 1.0f / 65536.0f; a.prox = to_coef(pp); a.u_first = 1;
    hipError_t e = launch_frows<false, 0, true, false>(s, np, a);
    for (int i = 0; i < iters && e == hipSuccess; ++i) {
        e = launch_fcols(f, s, pair0, np, dc_c);
        if (e != hipSuccess) break;
        const bool last = (i == iters - 1);
        a.u_first = (i == 0);
        if (prox == 2)      e = last ? launch_frows<true, 2, false, true>(s, np, a) : launch_frows<true, 2, true, false>(s, np, a);
        else if (prox == 1) e = last ? launch_frows<true, 1, false, true>(s, np, a) : launch_frows<true, 1, true, false>(s, np, a);
        else                e = last ? launch_frows<true, 3, false, true>(s, np, a) : launch_frows<true, 3, true, false>(s, np, a);
    }
    return e;
}

static FRowArgs row_args(Fused256* f, float* z, float* w, float* x, int c0, int Bc, const ProxParams& pp) {
    const size_t so = (size_t)c0 * 65536;
    FRowArgs a;
    a.T = f->T + (size_t)(c0 / 2) * 65536;
    a.z_in = z + so; a.w_in = w + so; a.z_out = z + so; a.w_out = w + so; a.x_out = x + so; a.B = Bc;
    a.scale = 1.0f / 65536.0f; a.prox = to_coef(pp); a.u_first = 1;
    return a;
}

template <bool HAS_INV, int PROX, bool HAS_FWD, bool WRITE_X>
static hipError_t launch_mixed_t(hipStream_t s, const FRowArgs& ra, int npR, const FColArgs& ca, int npC) {
    const int nR = npR * 16, nC = npC * 9;
    hipLaunchKernelGGL((k_fmixed<HAS_INV, PROX, HAS_FWD, WRITE_X>), dim3(nR + nC), dim3(256), 0, s, ra, ca, nR, nC);
    return hipGetLastError();
}

// kind: 0 = first (forward only), 1 = mid, 2 = last;  prox: 1 L1 two-state, 2 CNC, 3 L1 single-state
static hipError_t launch_mixed(hipStream_t s, int kind, int prox, const FRowArgs& ra, int npR, const FColArgs& ca, int npC) {
    if (kind == 0) return launch_mixed_t<false, 0, true, false>(s, ra, npR, ca, npC);
    if (kind == 1) {
        if (prox == 2) return launch_mixed_t<true, 2, true, false>(s, ra, npR, ca, npC);
        if (prox == 3) return launch_mixed_t<true, 3, true, false>(s, ra, npR, ca, npC);
        return launch_mixed_t<true, 1, true, false>(s, ra, npR, ca, npC);
    }
    if (prox == 2) return launch_mixed_t<true, 2, false, true>(s, ra, npR, ca, npC);
    if (prox == 3) return launch_mixed_t<true, 3, false, true>(s, ra, npR, ca, npC);
    return launch_mixed_t<true, 1, false, true>(s, ra, npR, ca, npC);
}
static hipError_t launch_rows_kind(hipStream_t s, int kind, int prox, const FRowArgs& ra, int np) {
    if (kind == 0) return launch_frows<false, 0, true, false>(s, np, ra);
    if (kind == 1) {
        if (prox == 2) return launch_frows<true, 2, true, false>(s, np, ra);
        if (prox == 3) return launch_frows<true, 3, true, false>(s, np, ra);
        return launch_frows<true, 1, true, false>(s, np, ra);
    }
    if (prox == 2) return launch_frows<true, 2, false, true>(s, np, ra);
    if (prox == 3) return launch_frows<true, 3, false, true>(s, np, ra);
    return launch_frows<true, 1, false, true>(s, np, ra);
}

// Staggered schedule over two halves A, B of the batch (slices are independent):
//   F(A) | C0(A)+F(B) | R0(A)+C0(B) | C1(A)+R0(B) | ... | R_{K-1}(A)+C_{K-1}(B) | R_{K-1}(B)
// every '+' is ONE mixed launch (k_fmixed): row workgroups of one half next to column workgroups
// of the other half on every CU.  Same arithmetic per slice as the sequential schedule: bit-identical.
static hipError_t run_mixed(Fused256* f, hipStream_t s, float* z, float* w, float* x, int c0, int B, int iters,
                            int prox, float dc_c, const ProxParams& pp) {
    const int BA = ((B / 2) + 1) & ~1, BB = B - BA;
    const int npA = BA / 2, npB = (BB + 1) / 2;
    FRowArgs ra = row_args(f, z, w, x, c0, BA, pp), rb = row_args(f, z, w, x, c0 + BA, BB, pp);
    const FColArgs ca = col_args(f, c0 / 2, dc_c), cb = col_args(f, c0 / 2 + npA, dc_c);
    hipError_t e = launch_rows_kind(s, 0, prox, ra, npA);                         // F(A)
    if (e == hipSuccess) e = launch_mixed(s, 0, prox, rb, npB, ca, npA);           // C0(A) + F(B)
    for (int i = 0; i < iters && e == hipSuccess; ++i) {
        const bool last = (i == iters - 1);
        ra.u_first = rb.u_first = (i == 0);
        e = launch_mixed(s, last ? 2 : 1, prox, ra, npA, cb, npB);                 // R_i(A) + C_i(B)
        if (e != hipSuccess) break;
        if (!last) e = launch_mixed(s, 1, prox, rb, npB, ca, npA);                 // C_{i+1}(A) + R_i(B)
        else       e = launch_rows_kind(s, 2, prox, rb, npB);                      // R_{K-1}(B)
    }
    return e;
}

// Slices are independent, so the K-iteration chains of different parts of the batch may run on
// different queues, share launches (run_mixed) or run one chunk after another (chunk*(z,w,T,Yh)
// <= the 256 MiB Infinity Cache keeps a chunk's working set on die; measured +-2 %).
hipError_t fused256_run(Fused256* f, hipStream_t s, float* z, float* w, float* x, int B, int iters, bool cnc,
                        float dc_c, ProxParams pp, const FusedSchedule& sch) {
    if (iters <= 0) return hipSuccess;
    const int prox = cnc ? 2 : (sch.l1_two_state ? 1 : 3);
    int queues = sch.queues < 1 ? 1 : (sch.queues > Fused256::MAXQ ? Fused256::MAXQ : sch.queues);
    if (sch.chunk > 0) queues = 1;
    if (queues == 1 && sch.chunk <= 0 && sch.mixed && B >= 64) return run_mixed(f, s, z, w, x, 0, B, iters, prox, dc_c, pp);
    if (queues >= 2 && B >= 32 * queues) {
        hipError_t e = hipSuccess;
        if (!f->ev_fork) e = hipEventCreateWithFlags(&f->ev_fork, hipEventDisableTiming);
        for (int q = 0; q < queues - 1 && e == hipSuccess; ++q) {
            if (f->side[q]) continue;
            e = hipStreamCreateWithFlags(&f->side[q], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_join[q], hipEventDisableTiming);
        }
        if (e != hipSuccess) return e;
        e = hipEventRecord(f->ev_fork, s);
        int c0 = 0;
        for (int q = 0; q < queues && e == hipSuccess; ++q) {
            const int Bq = (q == queues - 1) ? (B - c0) : (((B / queues) + 1) & ~1);     // even-sized parts
            hipStream_t sq = (q == 0) ? s : f->side[q - 1];
            if (q > 0) e = hipStreamWaitEvent(sq, f->ev_fork, 0);
            if (e == hipSuccess) e = (sch.mixed && Bq >= 64) ? run_mixed(f, sq, z, w, x, c0, Bq, iters, prox, dc_c, pp)
                                                              : run_chunk(f, sq, z, w, x, c0, Bq, iters, prox, dc_c, pp);
            if (q > 0 && e == hipSuccess) e = hipEventRecord(f->ev_join[q - 1], sq);
            if (q > 0 && e == hipSuccess) e = hipStreamWaitEvent(s, f->ev_join[q - 1], 0);
            c0 += Bq;
        }
        return e;
    }
    int chunk = sch.chunk > 0 ? (sch.chunk & ~1) : B;
    if (chunk < 2) chunk = 2;
    hipError_t e = hipSuccess;
    for (int c0 = 0; c0 < B && e == hipSuccess; c0 += chunk)
        e = run_chunk(f, s, z, w, x, c0, (B - c0 < chunk) ? (B - c0) : chunk, iters, prox, dc_c, pp);
    return e;
}

hipError_t fused256_dc(Fused256* f, hipStream_t s, const float* z, const float* w, float* x, int B, float dc_c) {
    const int np = (B + 1) / 2;
    FRowArgs a;
    a.T = f->T; a.z_in = z; a.w_in = w; a.z_out = nullptr; a.w_out = nullptr; a.x_out = x; a.B = B;
    a.scale = 1.0f / 65536.0f; a.prox = ProxCoef{}; a.u_first = 1;
    hipError_t e = launch_frows<false, 0, true, false>(s, np, a);
    if (e == hipSuccess) e = launch_fcols(f, s, 0, np, dc_c);
    if (e == hipSuccess) e = launch_frows<true, 0, false, true>(s, np, a);
    return e;
}


// ==========================================================================================
// "Split chain" column kernel and the engine built on it (float and double).
//
// The two-slice unpack commutes with the column transform (fft16.h, "split chains"), so each
// thread runs the whole column chain  FFT -> blend -> inverse FFT  of ONE slice on 16 values and
// the partner slice lives in the neighbouring lane (lane ^ 1, one DPP move per register):
//   * per thread 16 complex values instead of P[16] and Q[16]  -> half the data registers, and the
//     blend operands are one complex value per point instead of a float4: this is what makes a
//     double-precision instance fit, and it lifts the float instance to 4 waves per SIMD;
//   * lane s of a pair loads / stores the `s` half of each 2-element {C[r][k2], C[r][256-k2]}
//     group, so a wave instruction still covers whole 128-byte (float) / 256-byte (double) row
//     segments of 8 column pairs.
// ==========================================================================================
__device__ __forceinline__ float dpp_swap1(float v) {              // value of lane ^ 1
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
}
__device__ __forceinline__ double dpp_swap1(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffll), 0xB1, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), 0xB1, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <typename R> __device__ __forceinline__ cxT<R> dpp_swap1(cxT<R> v) { return mk<R>(dpp_swap1(v.x), dpp_swap1(v.y)); }

template <typename R>
struct FCol2Args {
    cxT<R>* T;
    const cxT<R>* Yh;
    const uint32_t* Mh;
    R c;
};

// One block per (tile of 8 columns k2 = 8 m + kl, slice pair); tile 16 = the self-mirrored columns k2 = 0 and 128.  As in
// k_fprepare each slice's y / mask tile and its mirror image (rows -k1, columns -k2) go through LDS: global memory is read in
// row segments of 8 complex values (64 bytes in float, 128 in double) instead of one element per 2 KiB (4 KiB) row, and the
// table is written in its own order.  (Until round 4 a block held ONE column: 256 threads x 2 KiB stride, the line fetched for
// every element.)  Arithmetic = hermitian_entry_t (fused_layout.h): unsampled entries are selected away, never multiplied.
constexpr int FP2_P = 9;
template <typename R>
__global__ __launch_bounds__(256) void k_fprepare2(const cxT<R>* y, const uint8_t* mask_bank, const int32_t* mask_id,
                                                   cxT<R>* Yh, uint32_t* Mh, int B) {
    using C = cxT<R>;
    __shared__ C yd[256 * FP2_P], ym[256 * FP2_P];               // direct tile [row][kl], mirror tile [row][kl] = y[row][-k2]
    __shared__ uint8_t md[256 * FP2_P], mm[256 * FP2_P];
    __shared__ uint8_t nib[256 * 8];                             // [k1][kl]: code of slice a | code of slice b << 2
    const int tid = threadIdx.x, m = blockIdx.x, pair = blockIdx.y;
    const bool self = (m == F2_TILES - 1);
    const int ncol = self ? 2 : 8;
    auto col_of = [&](int kl) { return self ? (kl ? 128 : 0) : 8 * m + kl; };
    for (int e = tid; e < 256 * 8; e += 256) nib[e] = 0;
#pragma unroll 1
    for (int sidx = 0; sidx < 2; ++sidx) {
        const int sl = 2 * pair + sidx;
        __syncthreads();
        if (sl < B) {
            const int mid = mask_id ? mask_id[sl] : 0;
            const C* ys = y + (size_t)sl * 65536;
            const uint8_t* ms = mask_bank + (size_t)mid * 65536;
#pragma unroll 4
            for (int i = 0; i < 8; ++i) {
                const int idx = tid + 256 * i, r = idx >> 3, c = idx & 7;
                if (c < ncol) {
                    const int k2 = col_of(c), k2m = (256 - k2) & 255;
                    yd[r * FP2_P + c] = ys[r * 256 + k2];
                    md[r * FP2_P + c] = ms[r * 256 + k2];
                    ym[r * FP2_P + c] = ys[r * 256 + k2m];
                    mm[r * FP2_P + c] = ms[r * 256 + k2m];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int o = tid + 256 * i;                         // [wave 4][j 16][tq 4][kl 8]: the lanes of this slice, in storage order
            const int kl = o & 7, tq = (o >> 3) & 3, j = (o >> 5) & 15, wv = o >> 9;
            const int k1 = 4 * wv + tq + 16 * j, r2 = (256 - k1) & 255;
            const bool valid = kl < ncol && (self || 8 * m + kl >= 1);      // k2 = 0 lives in tile 16
            if (!valid) continue;
            C yh = mk<R>((R)0, (R)0);
            int code = 0;
            if (sl < B) {
                const int m1 = md[k1 * FP2_P + kl] != 0, m2 = mm[r2 * FP2_P + kl] != 0;
                const C y1 = yd[k1 * FP2_P + kl], y2 = ym[r2 * FP2_P + kl];
                yh = mk<R>((R)0.5 * ((m1 ? y1.x : (R)0) + (m2 ? y2.x : (R)0)), (R)0.5 * ((m1 ? y1.y : (R)0) - (m2 ? y2.y : (R)0)));
                code = m1 + m2;
            }
            nib[k1 * 8 + kl] |= (uint8_t)(code << (2 * sidx));   // one thread per (k1, kl): no race
            Yh[yh2_index(pair, col_of(kl), j, 4 * wv + tq, sidx)] = yh;
        }
    }
    __syncthreads();
    {
        const int s2 = tid & 1, kl = (tid >> 1) & 7, t = tid >> 4;
        if (kl < ncol && (self || 8 * m + kl >= 1)) {
            uint32_t v = 0;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) v |= (uint32_t)((nib[(t + 16 * jj) * 8 + kl] >> (2 * s2)) & 3) << (2 * jj);
            Mh[mh2_index(pair, col_of(kl), t, s2)] = v;
        }
    }
}

template <typename R>
__global__ __launch_bounds__(256) void k_fcols2(FCol2Args<R> p) {
    using C = cxT<R>;
    __shared__ __attribute__((aligned(16))) C lds[COLS_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int s = lane & 1, kl = (lane >> 1) & 7, t = 4 * wv + (lane >> 4);
    const int pair = blockIdx.x / F2_TILES, m = blockIdx.x % F2_TILES;
    const bool self = (m == F2_TILES - 1);
    const bool valid = self ? (kl < 2) : (8 * m + kl >= 1);
    // element this lane moves: half `s` of the {column k2, column 256-k2} group (physical 2 k2 + s);
    // the self-mirrored columns 0 / 128 are single elements (physical 0 / 1) that both lanes read
    const int phys = self ? kl : 2 * (8 * m + kl) + s;
    C* Tp = p.T + (size_t)pair * 65536 + phys;
    C* twl = lds + 16 * CP;
    twl[17 * (tid >> 4) + (tid & 15)] = tw_table<R>()[((tid >> 4) * (tid & 15)) & 255];   // [t][k] = W256^(t k), rows of 17: conflict-free
    C a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = valid ? Tp[(t + 16 * j) * 256] : mk<R>((R)0, (R)0);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const C other = self ? a[j] : dpp_swap1(a[j]);
        const C pv = s ? other : a[j], qv = s ? a[j] : other;
        a[j] = s ? unpack_b(pv, qv) : unpack_a(pv, qv);          // this slice's row transform, column k2
    }
    C* region = lds + (lane & 15) * CP;
    __syncthreads();                             // twiddle table visible
    fft256_head_lds<false>(a, twl, t);
    col_exchange<false>(a, region, t);
    fft256_tail<false>(a);                       // a[j] = V_s[k1 = t + 16 j, k2]
    if (valid) {
        const size_t tb = (size_t)(m * 4 + wv);
        const uint32_t code = p.Mh[(size_t)pair * MH2_PAIR + tb * 64 + lane];
        const C* yhp = p.Yh + (size_t)pair * YH2_PAIR + tb * 16 * 64 + lane;
        const R ch = (R)0.5 * p.c;
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = blend_one(a[j], yhp[j * 64], (int)((code >> (2 * j)) & 3u), p.c, ch);
    }
    fft256_head_lds<true>(a, twl, t);
    col_exchange<true>(a, region, t);
    fft256_tail<true>(a);                        // column k2 of this slice's blended field
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const C other = dpp_swap1(a[j]);
        const C xa = s ? other : a[j], xb = s ? a[j] : other;
        const C o = s ? repack_q(xa, xb) : repack_p(xa, xb);
        if (valid && !(self && s)) Tp[(t + 16 * j) * 256] = o;
    }
}

// engine on the split-chain kernels: sequential rows / columns launches over `queues` HIP queues
template <typename R>
struct Fused256S {
    int Bmax = 0, np = 0;
    cxT<R>* T = nullptr;
    cxT<R>* Yh = nullptr;
    uint32_t* Mh = nullptr;
    static constexpr int MAXQ2 = 4;
    hipStream_t side[MAXQ2] = {};             // further queues; [0] unused
    hipEvent_t ev_fork = nullptr, ev_join[MAXQ2] = {};
};

template <typename R>
void fused256s_destroy(Fused256S<R>* f) {
    if (!f) return;
    if (f->T) (void)hipFree(f->T);
    if (f->Yh) (void)hipFree(f->Yh);
    if (f->Mh) (void)hipFree(f->Mh);
    for (int q = 1; q < Fused256S<R>::MAXQ2; ++q) {
        if (f->side[q]) (void)hipStreamDestroy(f->side[q]);
        if (f->ev_join[q]) (void)hipEventDestroy(f->ev_join[q]);
    }
    if (f->ev_fork) (void)hipEventDestroy(f->ev_fork);
    delete f;
}

template <typename R>
Fused256S<R>* fused256s_create(int Bmax, hipError_t* err) {
    Fused256S<R>* f = new Fused256S<R>();
    f->Bmax = Bmax;
    f->np = (Bmax + 1) / 2;
    hipError_t e = hipMalloc((void**)&f->T, (size_t)f->np * 65536 * sizeof(cxT<R>));
    if (e == hipSuccess) e = hipMalloc((void**)&f->Yh, (size_t)f->np * YH2_PAIR * sizeof(cxT<R>));
    if (e == hipSuccess) e = hipMalloc((void**)&f->Mh, (size_t)f->np * MH2_PAIR * sizeof(uint32_t));
    if (e == hipSuccess) {
        static thread_local c32 hf[256];
        static thread_local c64 hd[256];
        for (int m = 0; m < 256; ++m) {
            const double a = -2.0 * M_PI * (double)m / 256.0;
            hd[m] = mk<double>(cos(a), sin(a));
            hf[m] = mk<float>((float)cos(a), (float)sin(a));
        }
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_twf), hf, sizeof(hf));
        if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_twd), hd, sizeof(hd));
    }
    if (e != hipSuccess) {
        fused256s_destroy(f);
        *err = e;
        return nullptr;
    }
    *err = hipSuccess;
    return f;
}

template <typename R>
hipError_t fused256s_prepare(Fused256S<R>* f, hipStream_t s, const void* y, const uint8_t* mask_bank,
                             const int32_t* mask_id, int B) {
    if (B > f->Bmax) return hipErrorInvalidValue;
    const int np = (B + 1) / 2;
    hipLaunchKernelGGL(k_fprepare2<R>, dim3(F2_TILES, np), dim3(256), 0, s, reinterpret_cast<const cxT<R>*>(y), mask_bank,
                       mask_id, f->Yh, f->Mh, B);
    return hipGetLastError();
}

template <typename R, bool HAS_INV, int PROX, bool HAS_FWD, bool WRITE_X>
static hipError_t launch_frows_t(hipStream_t s, int np, const FRowArgsT<R>& a) {
    hipLaunchKernelGGL((k_frows<R, HAS_INV, PROX, HAS_FWD, WRITE_X>), dim3(np * 16), dim3(256), 0, s, a);
    return hipGetLastError();
}
template <typename R>
static hipError_t launch_fcols2(Fused256S<R>* f, hipStream_t s, int pair0, int np, R c) {
    FCol2Args<R> a;
    a.T = f->T + (size_t)pair0 * 65536;
    a.Yh = f->Yh + (size_t)pair0 * YH2_PAIR;
    a.Mh = f->Mh + (size_t)pair0 * MH2_PAIR;
    a.c = c;
    hipLaunchKernelGGL(k_fcols2<R>, dim3(np * F2_TILES), dim3(256), 0, s, a);
    return hipGetLastError();
}

template <typename R>
static ProxCoefT<R> to_coef_t(const ProxParamsT<R>& p) {
    ProxCoefT<R> c;
    c.thr = p.thr; c.c1 = p.c1; c.c2 = p.c2; c.c3 = p.c3; c.ib = p.ib;
    return c;
}

// K iterations on slices [c0, c0 + Bc) enqueued on stream s
template <typename R>
static hipError_t run_chunk2(Fused256S<R>* f, hipStream_t s, R* z, R* w, R* x, int c0, int Bc, int iters, int prox, R dc_c,
                             const ProxParamsT<R>& pp) {
    const int np = (Bc + 1) / 2, pair0 = c0 / 2;
    const size_t so = (size_t)c0 * 65536;
    FRowArgsT<R> a;
    a.T = f->T + (size_t)pair0 * 65536;
    a.z_in = z + so; a.w_in = w + so; a.z_out = z + so; a.w_out = w + so; a.x_out = x + so; a.B = Bc;
    a.scale = (R)(1.0 / 65536.0); a.prox = to_coef_t<R>(pp); a.u_first = 1;
    hipError_t e = launch_frows_t<R, false, 0, true, false>(s, np, a);
    for (int i = 0; i < iters && e == hipSuccess; ++i) {
        e = launch_fcols2<R>(f, s, pair0, np, dc_c);
        if (e != hipSuccess) break;
        const bool last = (i == iters - 1);
        a.u_first = (i == 0);
        if (prox == 2)      e = last ? launch_frows_t<R, true, 2, false, true>(s, np, a) : launch_frows_t<R, true, 2, true, false>(s, np, a);
        else if (prox == 1) e = last ? launch_frows_t<R, true, 1, false, true>(s, np, a) : launch_frows_t<R, true, 1, true, false>(s, np, a);
        else                e = last ? launch_frows_t<R, true, 3, false, true>(s, np, a) : launch_frows_t<R, true, 3, true, false>(s, np, a);
    }
    return e;
}

template <typename R>
hipError_t fused256s_run(Fused256S<R>* f, hipStream_t s, R* z, R* w, R* x, int B, int iters, bool cnc, R dc_c,
                         ProxParamsT<R> pp, const FusedSchedule& sch) {
    if (iters <= 0) return hipSuccess;
    const int prox = cnc ? 2 : (sch.l1_two_state ? 1 : 3);
    // chunked round-robin schedule: internal.h, chunk_plan.  sch.chunk < 0 (PNP_FUSED_CHUNK=-1): two halves of the batch on
    // two queues (the round-1 schedule); sch.chunk_queues overrides the number of queues (experiment builds).
    const ChunkPlan plan = chunk_plan(B, sch, false, sizeof(R) == 8, sch.chunk_queues);
    const int Q = plan.queues, chunk = plan.chunk;
    hipError_t e = hipSuccess;
    if (Q < 2 || B <= chunk) {
        for (int c0 = 0; c0 < B && e == hipSuccess; c0 += chunk)
            e = run_chunk2<R>(f, s, z, w, x, c0, (B - c0 < chunk) ? (B - c0) : chunk, iters, prox, dc_c, pp);
        return e;
    }
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
        e = run_chunk2<R>(f, q ? f->side[q] : s, z, w, x, c0, (B - c0 < chunk) ? (B - c0) : chunk, iters, prox, dc_c, pp);
    }
    for (int q = 1; q < Q && e == hipSuccess; ++q) {
        e = hipEventRecord(f->ev_join[q], f->side[q]);
        if (e == hipSuccess) e = hipStreamWaitEvent(s, f->ev_join[q], 0);
    }
    return e;
}

template <typename R>
hipError_t fused256s_dc(Fused256S<R>* f, hipStream_t s, const R* z, const R* w, R* x, int B, R dc_c) {
    const int np = (B + 1) / 2;
    FRowArgsT<R> a;
    a.T = f->T; a.z_in = z; a.w_in = w; a.z_out = nullptr; a.w_out = nullptr; a.x_out = x; a.B = B;
    a.scale = (R)(1.0 / 65536.0); a.prox = ProxCoefT<R>{}; a.u_first = 1;
    hipError_t e = launch_frows_t<R, false, 0, true, false>(s, np, a);
    if (e == hipSuccess) e = launch_fcols2<R>(f, s, 0, np, dc_c);
    if (e == hipSuccess) e = launch_frows_t<R, true, 0, false, true>(s, np, a);
    return e;
}

#define PNP_INSTANTIATE_F256S(R)                                                                                     \
    template Fused256S<R>* fused256s_create<R>(int, hipError_t*);                                                   \
    template void fused256s_destroy<R>(Fused256S<R>*);                                                               \
    template hipError_t fused256s_prepare<R>(Fused256S<R>*, hipStream_t, const void*, const uint8_t*, const int32_t*, int); \
    template hipError_t fused256s_run<R>(Fused256S<R>*, hipStream_t, R*, R*, R*, int, int, bool, R, ProxParamsT<R>, const FusedSchedule&); \
    template hipError_t fused256s_dc<R>(Fused256S<R>*, hipStream_t, const R*, const R*, R*, int, R);
PNP_INSTANTIATE_F256S(float)
PNP_INSTANTIATE_F256S(double)


}  // namespace pnp
