// Pointwise phase shared by the fused row kernels (256x256 and 512x512): for 4 consecutive pixels
// of a slice pair, turn the inverse transform's output into x = |Re|, |Im| / N, run the z / w
// update of the selected solver on both slices with 16-byte global accesses, and leave
// v = z - w (both slices, interleaved as one complex value) for the forward transform.
//   S4:124 (abs), S1:123-126 / S4:127-132 (prox + dual), S4:119 (z - w)
#pragma once
#include "fft16.h"

namespace pnp {

template <typename R>
struct FRowArgsT {
    cxT<R>* T;
    const R* z_in;
    const R* w_in;
    R* z_out;
    R* w_out;
    R* x_out;
    int B;
    R scale;            // 1 / (H W)
    ProxCoefT<R> prox;
    int u_first;        // PROX 3: 1 while the w buffer still holds a genuine w
};
using FRowArgs = FRowArgsT<float>;

// four consecutive reals moved as 16-byte accesses (one for float, two for double)
template <typename R>
struct alignas(16) vec4T {
    R x, y, z, w;
};
template <typename R> __device__ __forceinline__ vec4T<R> mk4(R x, R y, R z, R w) { vec4T<R> v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }
__device__ __forceinline__ float abs_(float a) { return __builtin_fabsf(a); }
__device__ __forceinline__ double abs_(double a) { return __builtin_fabs(a); }

// PROX: 0 none, 1 L1, 2 CNC, 3 L1 in single-state form: for L1, z = soft(u) and w = u - z are both
// functions of u = x + w_old, so between the first and the last iteration of a run only u is kept
// (in the w buffer): 8 N instead of 16 N state bytes per slice-iteration.  w_old is recomputed as
// u - soft(u) -- the very expression that produces the stored w -- so results are bit-identical
// to the two-state form.
// cell: the 4 complex LDS values of these pixels (x in, v out); offa / offb: element offsets of the
// pixels in slice a / b of the [B][H][W] float arrays.
// Off: size_t in the two-launch kernels; the slice-resident kernel passes 32-bit lane offsets on
// wave-uniform base pointers so that every access is "scalar base + 32-bit lane offset".
template <bool HAS_INV, int PROX, bool HAS_FWD, bool WRITE_X, typename R, typename Off>
__device__ __forceinline__ void pointwise4(const FRowArgsT<R>& p, cxT<R>* cell, Off offa, Off offb, bool has_b) {
    using V4 = vec4T<R>;
    R xa[4] = {0, 0, 0, 0}, xb[4] = {0, 0, 0, 0};
    if (HAS_INV) {
        const V4 c01 = *reinterpret_cast<const V4*>(cell);
        const V4 c23 = *reinterpret_cast<const V4*>(cell + 2);
        xa[0] = abs_(c01.x) * p.scale; xb[0] = abs_(c01.y) * p.scale;
        xa[1] = abs_(c01.z) * p.scale; xb[1] = abs_(c01.w) * p.scale;
        xa[2] = abs_(c23.x) * p.scale; xb[2] = abs_(c23.y) * p.scale;
        xa[3] = abs_(c23.z) * p.scale; xb[3] = abs_(c23.w) * p.scale;
    }
    R za[4] = {0, 0, 0, 0}, wa[4] = {0, 0, 0, 0}, zb[4] = {0, 0, 0, 0}, wb[4] = {0, 0, 0, 0};
    if (PROX == 3) {
        const V4 q1 = *reinterpret_cast<const V4*>(p.w_in + offa);
        wa[0] = q1.x; wa[1] = q1.y; wa[2] = q1.z; wa[3] = q1.w;
        if (has_b) {
            const V4 q2 = *reinterpret_cast<const V4*>(p.w_in + offb);
            wb[0] = q2.x; wb[1] = q2.y; wb[2] = q2.z; wb[3] = q2.w;
        }
        R ua[4], ub[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!p.u_first) {                        // buffer holds u_old: w_old = u_old - soft(u_old)
                wa[q] = wa[q] - soft_thr(wa[q], p.prox.thr);
                wb[q] = wb[q] - soft_thr(wb[q], p.prox.thr);
            }
            ua[q] = xa[q] + wa[q];
            ub[q] = xb[q] + wb[q];
            za[q] = soft_thr(ua[q], p.prox.thr); wa[q] = ua[q] - za[q];
            zb[q] = soft_thr(ub[q], p.prox.thr); wb[q] = ub[q] - zb[q];
        }
        if (HAS_FWD) {                               // mid-run: keep only u
            *reinterpret_cast<V4*>(p.w_out + offa) = mk4<R>(ua[0], ua[1], ua[2], ua[3]);
            if (has_b) *reinterpret_cast<V4*>(p.w_out + offb) = mk4<R>(ub[0], ub[1], ub[2], ub[3]);
        } else {                                     // end of run: materialise z and w
            *reinterpret_cast<V4*>(p.z_out + offa) = mk4<R>(za[0], za[1], za[2], za[3]);
            *reinterpret_cast<V4*>(p.w_out + offa) = mk4<R>(wa[0], wa[1], wa[2], wa[3]);
            if (has_b) {
                *reinterpret_cast<V4*>(p.z_out + offb) = mk4<R>(zb[0], zb[1], zb[2], zb[3]);
                *reinterpret_cast<V4*>(p.w_out + offb) = mk4<R>(wb[0], wb[1], wb[2], wb[3]);
            }
        }
    }
    if ((PROX != 0 && PROX != 3) || !HAS_INV) {
        const V4 v1 = *reinterpret_cast<const V4*>(p.z_in + offa);
        const V4 v2 = *reinterpret_cast<const V4*>(p.w_in + offa);
        za[0] = v1.x; za[1] = v1.y; za[2] = v1.z; za[3] = v1.w;
        wa[0] = v2.x; wa[1] = v2.y; wa[2] = v2.z; wa[3] = v2.w;
        if (has_b) {
            const V4 v3 = *reinterpret_cast<const V4*>(p.z_in + offb);
            const V4 v4 = *reinterpret_cast<const V4*>(p.w_in + offb);
            zb[0] = v3.x; zb[1] = v3.y; zb[2] = v3.z; zb[3] = v3.w;
            wb[0] = v4.x; wb[1] = v4.y; wb[2] = v4.z; wb[3] = v4.w;
        }
    }
    if (PROX == 1 || PROX == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (PROX == 1) { prox_l1_pt(xa[q], za[q], wa[q], p.prox); prox_l1_pt(xb[q], zb[q], wb[q], p.prox); }
            else           { prox_cnc_pt(xa[q], za[q], wa[q], p.prox); prox_cnc_pt(xb[q], zb[q], wb[q], p.prox); }
        }
        *reinterpret_cast<V4*>(p.z_out + offa) = mk4<R>(za[0], za[1], za[2], za[3]);
        *reinterpret_cast<V4*>(p.w_out + offa) = mk4<R>(wa[0], wa[1], wa[2], wa[3]);
        if (has_b) {
            *reinterpret_cast<V4*>(p.z_out + offb) = mk4<R>(zb[0], zb[1], zb[2], zb[3]);
            *reinterpret_cast<V4*>(p.w_out + offb) = mk4<R>(wb[0], wb[1], wb[2], wb[3]);
        }
    }
    if (WRITE_X) {
        *reinterpret_cast<V4*>(p.x_out + offa) = mk4<R>(xa[0], xa[1], xa[2], xa[3]);
        if (has_b) *reinterpret_cast<V4*>(p.x_out + offb) = mk4<R>(xb[0], xb[1], xb[2], xb[3]);
    }
    if (HAS_FWD) {
        *reinterpret_cast<V4*>(cell) = mk4<R>(za[0] - wa[0], zb[0] - wb[0], za[1] - wa[1], zb[1] - wb[1]);
        *reinterpret_cast<V4*>(cell + 2) = mk4<R>(za[2] - wa[2], zb[2] - wb[2], za[3] - wa[3], zb[3] - wb[3]);
    }
}

}  // namespace pnp
