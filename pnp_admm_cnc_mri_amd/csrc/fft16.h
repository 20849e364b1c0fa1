// In-register 16-point DFT and the two halves of the 16-lane cooperative 256-point FFT used by
// the fused kernels.  __host__ __device__ so the index math is unit-tested under g++ on the CPU
// (tests/test_host_cores.py) exactly as it runs on gfx950.
//
// 256 = 16 x 16 Cooley-Tukey over 16 cooperating lanes (t = 0..15), 16 points per lane:
//   in : lane t holds x[t + 16 j],  j = 0..15
//   (1) dft16 over j            -> Y[t][k2]
//   (2) twiddle by W256^(t*k2)   (tw[k2], generated in fp64, conj for the inverse direction)
//   (3) 16x16 transpose between the lanes (through LDS): lane k2 receives Y[n1][k2], n1 = 0..15
//   (4) dft16 over n1           -> lane k2 holds X[k2 + 16 k1], k1 = 0..15
// so input and output have the same "stride-16" distribution, and a forward transform can feed
// a pointwise stage and an inverse transform with no re-ordering in between.
#pragma once

#if defined(__HIPCC__)
#define PNP_HD __host__ __device__ __forceinline__
#else
#define PNP_HD inline
#endif

namespace pnp {

// complex value in the real type R: float for the production path, double for the fp64 engine
template <typename R>
struct cxT {
    R x, y;
};
using c32 = cxT<float>;
using c64 = cxT<double>;

template <typename R> PNP_HD cxT<R> mk(R x, R y) { cxT<R> r; r.x = x; r.y = y; return r; }
template <typename R> PNP_HD cxT<R> operator+(cxT<R> a, cxT<R> b) { return mk<R>(a.x + b.x, a.y + b.y); }
template <typename R> PNP_HD cxT<R> operator-(cxT<R> a, cxT<R> b) { return mk<R>(a.x - b.x, a.y - b.y); }
// The library is compiled with -ffp-contract=off and every fused multiply-add is written out, so
// the rounding sequence is fixed by this source and identical in every kernel instantiation
// (bit-identical results whether a run is split into several calls or not) and on the host.
PNP_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PNP_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <typename R> PNP_HD cxT<R> mul(cxT<R> a, cxT<R> b) { return mk<R>(fma_(a.x, b.x, -(a.y * b.y)), fma_(a.x, b.y, a.y * b.x)); }
template <typename R> PNP_HD cxT<R> mulc(cxT<R> a, cxT<R> b) { return mk<R>(fma_(a.x, b.x, a.y * b.y), fma_(a.y, b.x, -(a.x * b.y))); }   // a*conj(b)
template <bool INV, typename R> PNP_HD cxT<R> tmul(cxT<R> a, cxT<R> w);      // below: float device code takes the packed form
// multiply by -i (forward) / +i (inverse)
template <bool INV, typename R> PNP_HD cxT<R> rot(cxT<R> a) { return INV ? mk<R>(-a.y, a.x) : mk<R>(a.y, -a.x); }

#if defined(__HIPCC__)
// ----------------------------------------------------------------------------------------------
// Packed-fp32 transform core (device code, float only).  The slice-resident kernel was bound by VALU issue (4 cycles
// per wave64 instruction), not by HBM, and a third of the transform's instructions were v_mov: hipcc
// forms "a +- i b" and the complex products with v_pk_* but assembles their halves with moves, because it never
// uses DIFFERENT negations for the two halves.  The four primitives below spell those instructions out (op_sel picks
// the half of each 64-bit source, neg_lo / neg_hi negate per half); rounding sequence = dft4 / mul / mulc of
// below, so results are bit-identical to the generic dft16<INV>.  One 16-lane FFT-256 pass: 282 -> 192 VALU instructions.
// dft16<INV, float> and tmul<INV, float> dispatch here in device code.
// ----------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 to2(c32 a) { f2 r; r.x = a.x; r.y = a.y; return r; }
__device__ __forceinline__ c32 from2(f2 a) { return mk<float>(a.x, a.y); }
__device__ __forceinline__ f2 k2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
// p = a + rot(d), m = a - rot(d);  rot = multiplication by -i (forward) / +i (inverse)
template <bool INV>
__device__ __forceinline__ void addsub_rot(f2 a, f2 d, f2& p, f2& m) {
    if (!INV) {
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(p) : "v"(a), "v"(d));
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(m) : "v"(a), "v"(d));
    } else {
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(p) : "v"(a), "v"(d));
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(m) : "v"(a), "v"(d));
    }
}
template <bool INV>
__device__ __forceinline__ f2 rot2(f2 a) {     // 1.0 * a with the halves swapped and one of them negated
    f2 r;
    if (!INV) asm("v_pk_mul_f32 %0, %1, 1.0 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(r) : "v"(a));
    else      asm("v_pk_mul_f32 %0, %1, 1.0 op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[1,0]" : "=v"(r) : "v"(a));
    return r;
}
// a * w (forward) / a * conj(w) (inverse):  t = (a.y w.y, a.y w.x | a.x w.y);  r = (a.x w.x -+ t.lo, a.x w.y + t.hi | a.y w.x - t.hi)
// Both instructions sit in ONE asm statement: hipcc's hazard recognizer cannot see into inline asm and assumes that any asm
// result may come from a dst_sel instruction (gfx940+ "dst_sel forwarding hazard"), so it put an s_nop between a product in
// one asm statement and its use in the next -- 236 wasted issue slots per wave and iteration of the slice kernel.  A plain
// VALU dependency needs no software wait state.
#define PNP_PK_TMUL(CONSTRAINT)                                                                                                           \
    f2 r;                       /* the product of the first instruction lives in the result register: one asm output only */             \
    if (!INV) {                                                                                                                           \
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]\n\t"                                                                     \
            "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]"                                                  \
            : "=&v"(r) : "v"(a), CONSTRAINT(w));                                                                                          \
    } else {                                                                                                                              \
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]\n\t"                                                                     \
            "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_hi:[0,0,1]"                                                  \
            : "=&v"(r) : "v"(a), CONSTRAINT(w));                                                                                          \
    }                                                                                                                                     \
    return r;
template <bool INV> __device__ __forceinline__ f2 tmul_v(f2 a, f2 w) { PNP_PK_TMUL("v") }      // w in registers (table twiddles)
template <bool INV> __device__ __forceinline__ f2 tmul_s(f2 a, f2 w) { PNP_PK_TMUL("s") }      // w a constant: scalar register pair
#undef PNP_PK_TMUL
template <bool INV>
__device__ __forceinline__ void dft4_pk(f2& a0, f2& a1, f2& a2, f2& a3) {
    const f2 t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, d = a1 - a3;
    a0 = t0 + t2;
    a2 = t0 - t2;
    addsub_rot<INV>(t1, d, a1, a3);
}
// dft16<INV> below, operation for operation
template <bool INV>
__device__ __forceinline__ void dft16_pk(f2 (&a)[16]) {
    const float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) dft4_pk<INV>(a[n0], a[n0 + 4], a[n0 + 8], a[n0 + 12]);
    a[1 + 4] = tmul_s<INV>(a[1 + 4], k2(C1, -S1));
    a[1 + 8] = tmul_s<INV>(a[1 + 8], k2(H, -H));
    a[1 + 12] = tmul_s<INV>(a[1 + 12], k2(S1, -C1));
    a[2 + 4] = tmul_s<INV>(a[2 + 4], k2(H, -H));
    a[2 + 8] = rot2<INV>(a[2 + 8]);
    a[2 + 12] = tmul_s<INV>(a[2 + 12], k2(-H, -H));
    a[3 + 4] = tmul_s<INV>(a[3 + 4], k2(S1, -C1));
    a[3 + 8] = tmul_s<INV>(a[3 + 8], k2(-H, -H));
    a[3 + 12] = tmul_s<INV>(a[3 + 12], k2(-C1, S1));
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft4_pk<INV>(a[4 * k1], a[4 * k1 + 1], a[4 * k1 + 2], a[4 * k1 + 3]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i + 1; j < 4; ++j) { const f2 tmp = a[4 * i + j]; a[4 * i + j] = a[4 * j + i]; a[4 * j + i] = tmp; }
}

#endif  // __HIPCC__

template <bool INV, typename R>
PNP_HD void dft4(cxT<R>& a0, cxT<R>& a1, cxT<R>& a2, cxT<R>& a3) {
    const cxT<R> t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = rot<INV>(a1 - a3);
    a0 = t0 + t2;
    a1 = t1 + t3;
    a2 = t0 - t2;
    a3 = t1 - t3;
}

template <typename A, typename B> struct same_type { static constexpr bool value = false; };
template <typename A> struct same_type<A, A> { static constexpr bool value = true; };
template <bool INV, typename R> PNP_HD cxT<R> tmul(cxT<R> a, cxT<R> w) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (same_type<R, float>::value) {
        // table twiddles (values in registers): the two-instruction packed product; literal constants stay with the compiler
        if (!(__builtin_constant_p(w.x) && __builtin_constant_p(w.y))) return from2(tmul_v<INV>(to2(a), to2(w)));
    }
#endif
    return INV ? mulc(a, w) : mul(a, w);
}

// the float constants below are the correctly rounded values of the double literals
static_assert((float)0.92387953251128674 == 0.92387953251128674f && (float)0.38268343236508977 == 0.38268343236508977f &&
              (float)0.70710678118654752 == 0.70710678118654752f, "constant rounding");

// a[k] <- sum_n a[n] W16^(nk)   (W16 = exp(-2 pi i/16); conjugated when INV), natural order.
template <bool INV, typename R>
PNP_HD void dft16(cxT<R> (&a)[16]) {
    const R C1 = (R)0.92387953251128674;   // cos(pi/8)
    const R S1 = (R)0.38268343236508977;   // sin(pi/8)
    const R H = (R)0.70710678118654752;    // sqrt(1/2)
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (same_type<R, float>::value) {
        f2 p[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) p[k] = to2(a[k]);
        dft16_pk<INV>(p);
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = from2(p[k]);
        return;
    }
#endif
    // step 1: for n0: DFT4 over n1 of a[n0 + 4 n1]  -> b[n0][k1] stored at a[n0 + 4 k1]
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) dft4<INV>(a[n0], a[n0 + 4], a[n0 + 8], a[n0 + 12]);
    // step 2: b[n0][k1] *= W16^(n0*k1)   (forward values; tmul conjugates for INV)
    a[1 + 4] = tmul<INV>(a[1 + 4], mk<R>(C1, -S1));     // W^1
    a[1 + 8] = tmul<INV>(a[1 + 8], mk<R>(H, -H));       // W^2
    a[1 + 12] = tmul<INV>(a[1 + 12], mk<R>(S1, -C1));   // W^3
    a[2 + 4] = tmul<INV>(a[2 + 4], mk<R>(H, -H));       // W^2
    a[2 + 8] = rot<INV>(a[2 + 8]);                      // W^4 = -i
    a[2 + 12] = tmul<INV>(a[2 + 12], mk<R>(-H, -H));    // W^6
    a[3 + 4] = tmul<INV>(a[3 + 4], mk<R>(S1, -C1));     // W^3
    a[3 + 8] = tmul<INV>(a[3 + 8], mk<R>(-H, -H));      // W^6
    a[3 + 12] = tmul<INV>(a[3 + 12], mk<R>(-C1, S1));   // W^9
    // step 3: for k1: DFT4 over n0 of b[n0][k1] -> X[k1 + 4 k0]; b[n0][k1] sits at a[n0 + 4 k1]
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft4<INV>(a[4 * k1], a[4 * k1 + 1], a[4 * k1 + 2], a[4 * k1 + 3]);
    // now a[4 k1 + k0] = X[k1 + 4 k0]: transpose the 4x4 index to natural order
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i + 1; j < 4; ++j) {
            const cxT<R> tmp = a[4 * i + j];
            a[4 * i + j] = a[4 * j + i];
            a[4 * j + i] = tmp;
        }
}

// steps (1)+(2) for one lane; tw[k] = W256^(t*k) (forward values)
template <bool INV, typename R>
PNP_HD void fft256_head(cxT<R> (&a)[16], const cxT<R> (&tw)[16]) {
    dft16<INV>(a);
#pragma unroll
    for (int k = 1; k < 16; ++k) a[k] = tmul<INV>(a[k], tw[k]);
}
// step (4)
template <bool INV, typename R>
PNP_HD void fft256_tail(cxT<R> (&a)[16]) { dft16<INV>(a); }

// ----------------------------------------------------------------------------------------------
// 512 = 16 points x 32 lanes.  Two lane layouts:
//   t-layout : lane t (0..31), register j  <->  index t + 32 j
//   k-layout : lane L = 2 k2 + h (k2 0..15, h 0..1), register q  <->  index k2 + 16 q + 256 h
// "A" structure (t-layout in, k-layout out):  dft16 -> W512^(t k2) -> exchange -> dft16 -> W32^(q h)
//                                              -> radix-2 butterfly between lanes L and L^1
// "B" structure (k-layout in, t-layout out) is its transposed flow graph (the DFT matrix is
// symmetric):  butterfly -> W32^(q h) -> dft16 -> exchange -> W512^(t k2) -> dft16.
// INV conjugates every twiddle, i.e. selects the direction of the transform; both structures
// serve both directions.  twt = the lane's row of the W512^(t k) table (forward values); W32^(q h) comes from literals.
// ----------------------------------------------------------------------------------------------
// twt: the lane's 16 twiddles, twt[k] = W512^(t k) (a per-lane row of a [32][16] table: one address
// register + immediate offsets instead of 15 computed addresses)
template <bool INV>
PNP_HD void fft512_a1(c32 (&a)[16], const c32* twt) {                  // before the exchange
    dft16<INV>(a);
#pragma unroll
    for (int k = 1; k < 16; ++k) a[k] = tmul<INV>(a[k], twt[k]);
}
// W32^q = exp(-2 pi i q / 32) as compile-time constants (instruction literals: no table read, no
// register held); the h = 0 lane multiplies by 1.
PNP_HD c32 w32_or_one(int q, int h) {
    constexpr float C[16] = {1.0f, 0.980785251f, 0.923879504f, 0.831469595f, 0.707106769f, 0.555570245f, 0.382683426f, 0.195090324f, 6.12323426e-17f, -0.195090324f, -0.382683426f, -0.555570245f, -0.707106769f, -0.831469595f, -0.923879504f, -0.980785251f};
    constexpr float S[16] = {0.0f, 0.195090324f, 0.382683426f, 0.555570245f, 0.707106769f, 0.831469595f, 0.923879504f, 0.980785251f, 1.0f, 0.980785251f, 0.923879504f, 0.831469595f, 0.707106769f, 0.555570245f, 0.382683426f, 0.195090324f};
    return mk(h ? C[q] : 1.0f, h ? -S[q] : 0.0f);
}
template <bool INV>
PNP_HD void fft512_a2(c32 (&a)[16], int h) {                          // after it, before the butterfly
    dft16<INV>(a);
#pragma unroll
    for (int q = 1; q < 16; ++q) a[q] = tmul<INV>(a[q], w32_or_one(q, h));
}
// butterfly of lanes (k2,0),(k2,1): the h = 0 lane keeps the sum, the h = 1 lane the difference
PNP_HD c32 bfly2(c32 own, c32 other, int h) { return h ? (other - own) : (own + other); }
template <bool INV>
PNP_HD void fft512_b1(c32 (&a)[16], int h) {                          // after the butterfly, before the exchange
#pragma unroll
    for (int q = 1; q < 16; ++q) a[q] = tmul<INV>(a[q], w32_or_one(q, h));
    dft16<INV>(a);
}
template <bool INV>
PNP_HD void fft512_b2(c32 (&a)[16], const c32* twt) {                  // after the exchange
#pragma unroll
    for (int k = 1; k < 16; ++k) a[k] = tmul<INV>(a[k], twt[k]);
    dft16<INV>(a);
}

// ----------------------------------------------------------------------------------------------
// z / w updates shared by all kernels (S1:123-126, S4:127-132)
// ----------------------------------------------------------------------------------------------
template <typename R>
struct ProxCoefT {
    R thr, c1, c2, c3, ib;
};
using ProxCoef = ProxCoefT<float>;

// clamp(a, -c, c), c >= 0; one v_med3_f32 in float device code
template <typename R> PNP_HD R clamp_sym(R a, R c) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (same_type<R, float>::value) return __builtin_amdgcn_fmed3f(a, -c, c);
#endif
    return a < -c ? -c : (a > c ? c : a);
}
// soft(a, c) = max(|a| - c, 0) sign(a) (S1:18-19).  Float device code computes it as a - clamp(a, -c, c): the same value bit
// for bit (|a| <= c: a - a = 0; beyond: the very same subtraction), a zero's sign aside (tests/test_kernel_identities.py) --
// two instructions instead of a compare / select chain.
template <typename R> PNP_HD R soft_thr(R a, R c) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (same_type<R, float>::value) return a - clamp_sym(a, c);
#endif
    const R m = (a < 0 ? -a : a) - c;
    const R r = m > 0 ? m : (R)0;
    return a < 0 ? -r : r;
}
template <typename R> PNP_HD void prox_l1_pt(R x, R& z, R& w, const ProxCoefT<R>& p) {
    const R u = x + w;
    z = soft_thr(u, p.thr);
    w = u - z;
}
template <typename R> PNP_HD void prox_cnc_pt(R x, R& z, R& w, const ProxCoefT<R>& p) {
    const R u = x + w;
    const R cz = clamp_sym(z, p.ib);                           // z - soft(z, 1/b)
    const R t = fma_(p.c1, z, fma_(p.c2, u, p.c3 * cz));
    z = soft_thr(t, p.thr);
    w = u - z;
}

// ----------------------------------------------------------------------------------------------
// two-slice k-space blend of the fused column kernel.
//   P = C[k],  Q = C[-k]  with  C = F(va + i vb)  (va, vb real slices)
//   Va = (P + conj Q)/2, Vb = (P - conj Q)/(2i)
//   Xs = Vs (1 - c Mh_s) + c Yh_s      (Hermitian-symmetrised data-consistency blend)
//   P' = Xa + i Xb,  Q' = conj(Xa) + i conj(Xb)
// ma, mb in {0,1,2} = 2*Mh;  ch = c/2.
// ----------------------------------------------------------------------------------------------
PNP_HD void blend_pair(c32& P, c32& Q, c32 yha, c32 yhb, int ma, int mb, float c, float ch) {
    const float sx = P.x + Q.x, sy = P.y - Q.y;     // P + conj Q
    const float dx = P.x - Q.x, dy = P.y + Q.y;     // P - conj Q
    const float Aa = fma_(-0.5f * ch, (float)ma, 0.5f);  // (1 - c*Mh_a)/2
    const float Ab = fma_(-0.5f * ch, (float)mb, 0.5f);
    const float xax = fma_(Aa, sx, c * yha.x), xay = fma_(Aa, sy, c * yha.y);
    const float xbx = fma_(Ab, dy, c * yhb.x), xby = fma_(-Ab, dx, c * yhb.y);
    P = mk(xax - xby, xay + xbx);
    Q = mk(xax + xby, xbx - xay);
}


// ----------------------------------------------------------------------------------------------
// The same blend written per slice ("split chains", kernels_fused256.hip k_fcols2).  With
//   p[r] = T[r][k2],  q[r] = T[r][-k2]   (row-transformed field of the pair, one column and its mirror)
// the row transforms of the two REAL slices are  Ta[r][k2] = (p + conj q)/2,  Tb[r][k2] = (p - conj q)/(2i)
// (a real signal's transform is Hermitian), so the unpack can be done BEFORE the column transform
// and each slice runs its own chain  column FFT -> blend -> inverse column FFT:
//   X = V (1 - c Mh) + c Yh,   code = 2 Mh in {0,1,2},  ch = c/2.
// The blended field is again Hermitian in k2:  p' = xa' + i xb',  q' = conj(xa') + i conj(xb').
// ----------------------------------------------------------------------------------------------
template <typename R> PNP_HD cxT<R> unpack_a(cxT<R> p, cxT<R> q) { return mk<R>((R)0.5 * (p.x + q.x), (R)0.5 * (p.y - q.y)); }
template <typename R> PNP_HD cxT<R> unpack_b(cxT<R> p, cxT<R> q) { return mk<R>((R)0.5 * (p.y + q.y), (R)0.5 * (q.x - p.x)); }
template <typename R> PNP_HD cxT<R> blend_one(cxT<R> V, cxT<R> yh, int code, R c, R ch) {
    const R A = fma_(-ch, (R)code, (R)1);
    return mk<R>(fma_(A, V.x, c * yh.x), fma_(A, V.y, c * yh.y));
}
// blend_one with every coefficient scaled by a power of two s:  cs = c s, chs = ch s, os = s.  A_s = fma(-chs, code, os) is
// exactly s A and fma(A_s, V, cs yh) exactly s blend_one(V, ...) (powers of two commute with rounding, underflow aside:
// |values| < 2^-126 / s, i.e. 1e-33 for s = 2^-16 on a field of magnitude 1e4).  The slice-resident kernel uses it twice over:
//   s = 1/N   -- the inverse transforms' normalisation rides on the blend, x = |.| needs no multiplication;
//   s = 1/2N with the DOUBLED field V2 = 2 V its transposition leaves (the 1/2 of the real-to-complex unpack left out):
//             (A_s / 2)(2 V) = A_s V exactly.
template <typename R> PNP_HD cxT<R> blend_scaled_f(cxT<R> V, cxT<R> yh, R code, R cs, R chs, R os) {      // code already a float
    const R A = fma_(-chs, code, os);
    return mk<R>(fma_(A, V.x, cs * yh.x), fma_(A, V.y, cs * yh.y));
}
template <typename R> PNP_HD cxT<R> blend_scaled(cxT<R> V, cxT<R> yh, int code, R cs, R chs, R os) {
    return blend_scaled_f(V, yh, (R)code, cs, chs, os);
}
template <typename R> PNP_HD cxT<R> repack_p(cxT<R> xa, cxT<R> xb) { return mk<R>(xa.x - xb.y, xa.y + xb.x); }
template <typename R> PNP_HD cxT<R> repack_q(cxT<R> xa, cxT<R> xb) { return mk<R>(xa.x + xb.y, xb.x - xa.y); }

}  // namespace pnp

