// Generic c2c kernels of libpnpmri.so for gfx950: any H, W in {256, 512}.
//
// These implement, transform by transform, what np.fft.fft2 / np.fft.ifft2 and the surrounding
// NumPy lines of the reference do (S4:119-124 etc.; see include/pnp_mri.h).  They are the
// operator API (fft2, A, A^H, Df, synthesis, init) and the fallback of the iteration for shapes
// the fused 256x256 kernels (kernels_fused256.hip) do not cover.
//
// FFT: autosort (Stockham) radix-4 stages (+ one radix-2 stage for 512) in LDS, ping-pong
// buffers, twiddles from an fp64-generated fp32 table W_N^m = exp(-2 pi i m / N) staged in LDS.
//   rows   : one wavefront (64 lanes) per row, 4 rows per 256-thread workgroup
//   columns: a tile of 16 columns x N rows per workgroup, transposed into LDS on load so the
//            same row routine runs on it (16 lanes per column); global accesses are 128-B row
//            segments; the LDS pitch N+1 keeps the transposing writes conflict-free.
#include "internal.h"
#include <math.h>

namespace pnp {

__device__ float2 g_tw256[256];
__device__ float2 g_tw512[512];

hipError_t upload_twiddles() {
    static thread_local float2 h[512];
    for (int N : {256, 512}) {
        for (int m = 0; m < N; ++m) {
            double a = -2.0 * M_PI * (double)m / (double)N;
            h[m] = make_float2((float)cos(a), (float)sin(a));
        }
        hipError_t e = (N == 256) ? hipMemcpyToSymbol(HIP_SYMBOL(g_tw256), h, sizeof(float2) * 256)
                                  : hipMemcpyToSymbol(HIP_SYMBOL(g_tw512), h, sizeof(float2) * 512);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {   // a * conj(b)
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -(a.x * b.y)));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// Length-N transform of the contiguous LDS array `a` by T cooperating threads (t in [0,T)),
// scratch `b`; all threads of the workgroup must call it (block barriers inside).
// Returns the buffer holding the result (natural order).
template <int N, int T, bool INV>
__device__ __forceinline__ float2* fft_lds(float2* a, float2* b, const float2* tw, int t) {
    static_assert(N == 256 || N == 512, "N");
    int Ns = 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        for (int j = t; j < N / 4; j += T) {
            const int k = j & (Ns - 1);
            const int ti = k * (N / (Ns * 4));
            float2 v0 = a[j], v1 = a[j + N / 4], v2 = a[j + N / 2], v3 = a[j + 3 * N / 4];
            if (s > 0) {
                const float2 w1 = tw[ti], w2 = tw[2 * ti], w3 = tw[3 * ti];
                if (INV) { v1 = cmulc(v1, w1); v2 = cmulc(v2, w2); v3 = cmulc(v3, w3); }
                else     { v1 = cmul(v1, w1);  v2 = cmul(v2, w2);  v3 = cmul(v3, w3); }
            }
            const float2 t0 = cadd(v0, v2), t1 = csub(v0, v2), t2 = cadd(v1, v3);
            const float2 d = csub(v1, v3);
            const float2 t3 = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);   // (+/-) i * d
            const int j0 = ((j - k) << 2) + k;
            b[j0] = cadd(t0, t2);
            b[j0 + Ns] = cadd(t1, t3);
            b[j0 + 2 * Ns] = csub(t0, t2);
            b[j0 + 3 * Ns] = csub(t1, t3);
        }
        __syncthreads();
        float2* tmp = a; a = b; b = tmp;
        Ns *= 4;
    }
    if (N == 512) {                                   // final radix-2 stage, Ns = 256
        for (int j = t; j < N / 2; j += T) {
            const int k = j & 255;
            float2 v0 = a[j], v1 = a[j + N / 2];
            v1 = INV ? cmulc(v1, tw[k]) : cmul(v1, tw[k]);
            const int j0 = ((j - k) << 1) + k;
            b[j0] = cadd(v0, v1);
            b[j0 + 256] = csub(v0, v1);
        }
        __syncthreads();
        float2* tmp = a; a = b; b = tmp;
    }
    return a;
}

__device__ __forceinline__ float soft(float a, float c) { return copysignf(fmaxf(fabsf(a) - c, 0.0f), a); }

__device__ __forceinline__ void prox_l1(float x, float& z, float& w, const ProxParams& p) {
    const float u = x + w;
    z = soft(u, p.thr);
    w = u - z;
}
__device__ __forceinline__ void prox_cnc(float x, float& z, float& w, const ProxParams& p) {
    const float u = x + w;
    const float clipz = fminf(fmaxf(z, -p.ib), p.ib);          // z - soft(z, 1/b)
    const float t = fmaf(p.c1, z, fmaf(p.c2, u, p.c3 * clipz));
    z = soft(t, p.thr);
    w = u - z;
}

// ------------------------------------------------------------------------------------------
// rows
// ------------------------------------------------------------------------------------------
template <int N, int IN, bool INV, int EPI>
__global__ __launch_bounds__(256) void k_rows(RowArgs p) {
    constexpr int T = 64, ROWS = 4;
    __shared__ float2 sA[ROWS * N];
    __shared__ float2 sB[ROWS * N];
    __shared__ float2 sTw[N];
    const int tid = threadIdx.x, lane = tid & 63, rw = tid >> 6;
    const float2* gtw = (N == 256) ? g_tw256 : g_tw512;
    for (int i = tid; i < N; i += 256) sTw[i] = gtw[i];
    const int row = blockIdx.x * ROWS + rw;                    // nrows is a multiple of ROWS
    const size_t base = (size_t)row * N;
    float2* a = sA + rw * N;
    float2* b = sB + rw * N;
#pragma unroll
    for (int i = 0; i < N / T; ++i) {
        const int n = lane + i * T;
        float2 v;
        if (IN == IN_COMPLEX) v = p.cin[base + n];
        else if (IN == IN_REAL) v = make_float2(p.rin0[base + n], 0.0f);
        else v = make_float2(p.rin0[base + n] - p.rin1[base + n], 0.0f);
        a[n] = v;
    }
    __syncthreads();
    float2* r = fft_lds<N, T, INV>(a, b, sTw, lane);
#pragma unroll
    for (int i = 0; i < N / T; ++i) {
        const int n = lane + i * T;
        const float2 v = r[n];
        if (EPI == EPI_COMPLEX) {
            p.cout[base + n] = make_float2(v.x * p.scale, v.y * p.scale);
        } else if (EPI == EPI_ABS_REAL) {
            p.x_out[base + n] = fabsf(v.x * p.scale);
        } else if (EPI == EPI_ABS_COMPLEX) {
            p.x_out[base + n] = sqrtf(v.x * v.x + v.y * v.y) * p.scale;
        } else {
            const float x = fabsf(v.x * p.scale);
            float z = p.z[base + n], w = p.w[base + n];
            if (EPI == EPI_L1) prox_l1(x, z, w, p.prox); else prox_cnc(x, z, w, p.prox);
            p.z[base + n] = z;
            p.w[base + n] = w;
            if (p.x_out) p.x_out[base + n] = x;
        }
    }
}

template <int N, int IN, bool INV, int EPI>
static hipError_t launch_rows_t(hipStream_t s, const RowArgs& a) {
    hipLaunchKernelGGL((k_rows<N, IN, INV, EPI>), dim3(a.nrows / 4), dim3(256), 0, s, a);
    return hipGetLastError();
}

template <int N>
static hipError_t launch_rows_n(hipStream_t s, RowIn in, bool inv, RowEpi epi, const RowArgs& a) {
    if (!inv && epi == EPI_COMPLEX) {
        if (in == IN_COMPLEX)   return launch_rows_t<N, IN_COMPLEX, false, EPI_COMPLEX>(s, a);
        if (in == IN_REAL)      return launch_rows_t<N, IN_REAL, false, EPI_COMPLEX>(s, a);
        if (in == IN_REAL_DIFF) return launch_rows_t<N, IN_REAL_DIFF, false, EPI_COMPLEX>(s, a);
    }
    if (inv && in == IN_COMPLEX) {
        switch (epi) {
            case EPI_COMPLEX:     return launch_rows_t<N, IN_COMPLEX, true, EPI_COMPLEX>(s, a);
            case EPI_ABS_REAL:    return launch_rows_t<N, IN_COMPLEX, true, EPI_ABS_REAL>(s, a);
            case EPI_ABS_COMPLEX: return launch_rows_t<N, IN_COMPLEX, true, EPI_ABS_COMPLEX>(s, a);
            case EPI_L1:          return launch_rows_t<N, IN_COMPLEX, true, EPI_L1>(s, a);
            case EPI_CNC:         return launch_rows_t<N, IN_COMPLEX, true, EPI_CNC>(s, a);
        }
    }
    return hipErrorInvalidValue;
}

hipError_t launch_rows(hipStream_t s, int W, RowIn in, bool inv, RowEpi epi, const RowArgs& a) {
    if (a.nrows % 4) return hipErrorInvalidValue;
    if (W == 256) return launch_rows_n<256>(s, in, inv, epi, a);
    if (W == 512) return launch_rows_n<512>(s, in, inv, epi, a);
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------
// columns: [optional forward] -> pointwise k-space op -> [optional inverse]
// ------------------------------------------------------------------------------------------
template <int N, bool PRE, int MID, bool POST>
__global__ __launch_bounds__(256) void k_cols(ColArgs p, int W) {
    constexpr int COLS = 16, T = 16, P = N + 1;
    extern __shared__ float2 smem[];
    float2* sA = smem;                  // [COLS][P]
    float2* sB = smem + COLS * P;
    float2* sTw = smem + 2 * COLS * P;
    const int tid = threadIdx.x;
    const float2* gtw = (N == 256) ? g_tw256 : g_tw512;
    for (int i = tid; i < N; i += 256) sTw[i] = gtw[i];
    const int tiles = W / COLS;
    const int b = blockIdx.x / tiles;
    const int k0 = (blockIdx.x % tiles) * COLS;
    const size_t sbase = (size_t)b * N * W;
    for (int idx = tid; idx < N * COLS; idx += 256) {
        const int r = idx >> 4, c = idx & 15;
        sA[c * P + r] = p.in[sbase + (size_t)r * W + k0 + c];
    }
    __syncthreads();
    const int c_f = tid >> 4, t_f = tid & 15;
    float2* cur = sA;
    float2* oth = sB;
    if (PRE) {
        float2* r = fft_lds<N, T, false>(sA + c_f * P, sB + c_f * P, sTw, t_f);
        if (r != sA + c_f * P) { cur = sB; oth = sA; }
    }
    if (MID != MID_NONE) {
        const int mid = p.mask_id ? p.mask_id[b] : 0;
        const uint8_t* mask = p.mask_bank + (size_t)mid * N * W;
        const float2* yb = p.y + ((MID == MID_MASK_ADD && !p.y_per_slice) ? 0 : sbase);
        for (int idx = tid; idx < N * COLS; idx += 256) {
            const int r = idx >> 4, c = idx & 15;
            const size_t g = (size_t)r * W + k0 + c;
            float2 X = cur[c * P + r];
            const bool m = mask[g] != 0;
            if (MID == MID_BLEND) {
                if (m) { const float2 yv = yb[g]; X.x = fmaf(yv.x - X.x, p.c, X.x); X.y = fmaf(yv.y - X.y, p.c, X.y); }
            } else if (MID == MID_MASK) {
                if (!m) X = make_float2(0.f, 0.f);
            } else if (MID == MID_RESID) {
                if (m) { const float2 yv = yb[g]; X.x -= yv.x; X.y -= yv.y; } else X = make_float2(0.f, 0.f);
            } else if (MID == MID_MASK_ADD) {
                const float2 nv = yb[g];
                X = m ? cadd(X, nv) : nv;
            }
            cur[c * P + r] = X;
        }
        __syncthreads();
    }
    if (POST) {
        float2* r = fft_lds<N, T, true>(cur + c_f * P, oth + c_f * P, sTw, t_f);
        if (r != cur + c_f * P) { float2* tmp = cur; cur = oth; oth = tmp; }
    }
    for (int idx = tid; idx < N * COLS; idx += 256) {
        const int r = idx >> 4, c = idx & 15;
        p.out[sbase + (size_t)r * W + k0 + c] = cur[c * P + r];
    }
}

template <int N, bool PRE, int MID, bool POST>
static hipError_t launch_cols_t(hipStream_t s, int W, const ColArgs& a) {
    const size_t lds = sizeof(float2) * (2 * 16 * (N + 1) + N);
    static bool attr_done = false;          // >64 KiB dynamic LDS needs the opt-in once per kernel
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)k_cols<N, PRE, MID, POST>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((k_cols<N, PRE, MID, POST>), dim3(a.B * (W / 16)), dim3(256), lds, s, a, W);
    return hipGetLastError();
}

template <int N>
static hipError_t launch_cols_n(hipStream_t s, int W, bool pre, ColMid mid, bool post, const ColArgs& a) {
    if (pre && !post && mid == MID_NONE)      return launch_cols_t<N, true, MID_NONE, false>(s, W, a);
    if (!pre && post && mid == MID_NONE)      return launch_cols_t<N, false, MID_NONE, true>(s, W, a);
    if (pre && post && mid == MID_BLEND)      return launch_cols_t<N, true, MID_BLEND, true>(s, W, a);
    if (pre && !post && mid == MID_MASK)      return launch_cols_t<N, true, MID_MASK, false>(s, W, a);
    if (!pre && post && mid == MID_MASK)      return launch_cols_t<N, false, MID_MASK, true>(s, W, a);
    if (pre && post && mid == MID_RESID)      return launch_cols_t<N, true, MID_RESID, true>(s, W, a);
    if (pre && !post && mid == MID_MASK_ADD)  return launch_cols_t<N, true, MID_MASK_ADD, false>(s, W, a);
    return hipErrorInvalidValue;
}

hipError_t launch_cols(hipStream_t s, int H, int W, bool pre, ColMid mid, bool post, const ColArgs& a) {
    if (W % 16) return hipErrorInvalidValue;
    if (H == 256) return launch_cols_n<256>(s, W, pre, mid, post, a);
    if (H == 512) return launch_cols_n<512>(s, W, pre, mid, post, a);
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------
// pointwise kernels on caller pointers (PnP path, S6:301-308): 4 floats per lane, grid-stride
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

template <bool CNC>
__global__ __launch_bounds__(256) void k_prox(const float4* x, float4* z, float4* w, ProxParams p, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 xv = x[i];
        float4 zv = z[i], wv = w[i];
        if (CNC) { prox_cnc(xv.x, zv.x, wv.x, p); prox_cnc(xv.y, zv.y, wv.y, p); prox_cnc(xv.z, zv.z, wv.z, p); prox_cnc(xv.w, zv.w, wv.w, p); }
        else     { prox_l1(xv.x, zv.x, wv.x, p);  prox_l1(xv.y, zv.y, wv.y, p);  prox_l1(xv.z, zv.z, wv.z, p);  prox_l1(xv.w, zv.w, wv.w, p); }
        z[i] = zv; w[i] = wv;
    }
}

__global__ __launch_bounds__(256) void k_combine(const float4* z, const float4* x, const float4* w, const float4* sd,
                                                 float4* t, float c1, float c2, float c3, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 zv = z[i], xv = x[i], wv = w[i], sv = sd[i];
        float4 o;
        // S6:301 evaluated left to right in float32 like the reference's torch expression
        o.x = c1 * zv.x + c2 * (xv.x + wv.x) + c3 * (zv.x - sv.x);
        o.y = c1 * zv.y + c2 * (xv.y + wv.y) + c3 * (zv.y - sv.y);
        o.z = c1 * zv.z + c2 * (xv.z + wv.z) + c3 * (zv.z - sv.z);
        o.w = c1 * zv.w + c2 * (xv.w + wv.w) + c3 * (zv.w - sv.w);
        t[i] = o;
    }
}

__global__ __launch_bounds__(256) void k_add(const float4* a, const float4* b, float4* o, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 av = a[i], bv = b[i];
        o[i] = make_float4(av.x + bv.x, av.y + bv.y, av.z + bv.z, av.w + bv.w);
    }
}

__global__ __launch_bounds__(256) void k_dual_clamp(float4* x, float4* z, float4* w, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 xv = x[i], zv = z[i], wv = w[i];
        wv.x = wv.x + xv.x - zv.x; wv.y = wv.y + xv.y - zv.y; wv.z = wv.z + xv.z - zv.z; wv.w = wv.w + xv.w - zv.w;
        x[i] = make_float4(clamp01(xv.x), clamp01(xv.y), clamp01(xv.z), clamp01(xv.w));
        z[i] = make_float4(clamp01(zv.x), clamp01(zv.y), clamp01(zv.z), clamp01(zv.w));
        w[i] = make_float4(clamp01(wv.x), clamp01(wv.y), clamp01(wv.z), clamp01(wv.w));
    }
}

static inline unsigned pw_grid(size_t n4) {
    size_t g = (n4 + 255) / 256;
    return (unsigned)(g < 2048 ? (g ? g : 1) : 2048);
}

hipError_t launch_prox(hipStream_t s, bool cnc, const float* x, float* z, float* w, ProxParams p, size_t n) {
    const size_t n4 = n / 4;
    if (cnc) hipLaunchKernelGGL(k_prox<true>, dim3(pw_grid(n4)), dim3(256), 0, s, (const float4*)x, (float4*)z, (float4*)w, p, n4);
    else     hipLaunchKernelGGL(k_prox<false>, dim3(pw_grid(n4)), dim3(256), 0, s, (const float4*)x, (float4*)z, (float4*)w, p, n4);
    return hipGetLastError();
}
hipError_t launch_combine(hipStream_t s, const float* z, const float* x, const float* w, const float* sd, float* t,
                          float c1, float c2, float c3, size_t n) {
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(k_combine, dim3(pw_grid(n4)), dim3(256), 0, s, (const float4*)z, (const float4*)x,
                       (const float4*)w, (const float4*)sd, (float4*)t, c1, c2, c3, n4);
    return hipGetLastError();
}
hipError_t launch_add(hipStream_t s, const float* a, const float* b, float* o, size_t n) {
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(k_add, dim3(pw_grid(n4)), dim3(256), 0, s, (const float4*)a, (const float4*)b, (float4*)o, n4);
    return hipGetLastError();
}
hipError_t launch_dual_clamp(hipStream_t s, float* x, float* z, float* w, size_t n) {
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(k_dual_clamp, dim3(pw_grid(n4)), dim3(256), 0, s, (float4*)x, (float4*)z, (float4*)w, n4);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// metrics: per slice sum((255 x - gt)^2) and sum(gt^2) in double        (utils_image.py:543-636)
// one workgroup per slice; wave shuffle reduction then LDS across the 4 waves
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_metrics(const float* x, const uint8_t* gt, double* acc, int N) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* xb = x + (size_t)b * N;
    const uint8_t* gb = gt + (size_t)b * N;
    double se = 0.0, sg = 0.0;
    for (int i = tid; i < N; i += 256) {
        const double g = (double)gb[i];
        const double d = (double)xb[i] * 255.0 - g;
        se += d * d;
        sg += g * g;
    }
    for (int o = 32; o > 0; o >>= 1) { se += __shfl_down(se, o); sg += __shfl_down(sg, o); }
    __shared__ double s1[4], s2[4];
    if ((tid & 63) == 0) { s1[tid >> 6] = se; s2[tid >> 6] = sg; }
    __syncthreads();
    if (tid == 0) { acc[2 * b] = s1[0] + s1[1] + s1[2] + s1[3]; acc[2 * b + 1] = s2[0] + s2[1] + s2[2] + s2[3]; }
}

hipError_t launch_metrics(hipStream_t s, const float* x, const uint8_t* gt, double* acc, int B, int N) {
    hipLaunchKernelGGL(k_metrics, dim3(B), dim3(256), 0, s, x, gt, acc, N);
    return hipGetLastError();
}

}  // namespace pnp
