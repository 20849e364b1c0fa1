// Slice-resident gfx950 kernel for 256x256 slices: a whole ADMM run with the slice ON the compute unit.
//
// The two-launch fused path (kernels_fused256.hip) is bound by HBM traffic, and 16 of its 36 N bytes
// per slice-iteration are the transposed field T going out after the row pass and coming back for
// the column pass.  A CU of MI355X has a 512 KiB vector register file and 160 KiB of LDS: one REAL
// 256x256 slice (256 KiB as 128 packed complex rows, or as 128 half-plane complex columns) fits in
// the registers of one workgroup -- 512 threads x 256 VGPRs = the whole register file of the CU, 128 of them
// per thread holding data (four "register sets" of 16 complex values) -- and LDS is large enough to turn
// rows into columns in two passes.  So here ONE workgroup owns ONE slice for all iterations of a
// run and T never exists in memory:
//
//   per iteration and slice:  z, w read + written (16 N bytes; 8 N for ADMM_L1's single-state form),
//                             Hermitian measurement table read (4 N)              = 20 N (12 N) bytes
//
//   rows(first)                 v = z - w, row pairs (2r, 2r+1) packed as one complex row, 16-lane FFT-256
//   repeat iters times:
//     T1  row form -> column form through LDS (2 passes), real-to-complex unpack on the way (slice_layout.h)
//     columns                   FFT-256 -> Hermitian blend against Yh / Mh -> inverse FFT-256
//                               (127 half-plane columns + the packed column {k2 = 0, k2 = 128})
//     T2  column form -> row form, complex-to-real repack on the way
//     rows                      inverse FFT-256 -> x = |re|, |im| / N -> L1 / CNC z-update + dual update
//                               -> v = z - w -> FFT-256          (S4:119-132; the last iteration also writes x)
//
// z and w live in HBM in the row transform's own thread order ("slice order", slice_layout.h: sl_state_index), so the
// z- / w-update runs on the registers the inverse transform left and feeds the forward transform directly; api.hip converts
// to and from the natural [256][256] order (k_state_order below) whenever anything else looks at the state.
//
// Row and column phases are wave-local (a 16-lane transform group never leaves its wave, each wave
// has its own LDS region), so the 8 waves of the workgroup drift apart and cover each other's HBM
// latency; only the two transpositions are workgroup barriers.  Same arithmetic cores as the fused
// path (fft16.h, fused_pointwise.h); index maps verified on the CPU by tests/host/slice_resident_emulation.cpp.
// Measurements, the road here and the dead ends: DESIGN.md section 4.1.
#include "internal.h"
#include "fused_layout.h"
#include "slice_layout.h"
#include "fused_pointwise.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#ifndef SLICE_EARLY
#define SLICE_EARLY 1       // set 0 of the row phase: its first z / w accesses go out between the two passes of T2 (0: at the start of the phase; +0.8 % at 100 steps)
#endif
#ifndef SLICE_PRIOSWAP
#define SLICE_PRIOSWAP 1    // the two waves of a SIMD trade issue priority between register sets of the wave-local phases (0: never)
#endif
#ifndef SLICE_PRIOMASK
#define SLICE_PRIOMASK 6    // bit s: the older wave of each SIMD pair (waves 0-3) has the higher priority during register set s
#endif
#ifndef SLICE_YH_AUX
#define SLICE_YH_AUX 0      // cache policy bits of the Hermitian-table loads (experiment knob: 2 = nt)
#endif
#ifndef SLICE_ST_AUX
#define SLICE_ST_AUX 0      // cache policy bits of the state stores (experiment knob)
#endif
#ifndef SLICE_PF
#define SLICE_PF 3          // of a set's 8 z / w accesses per lane: fetched ahead of the transforms (experiment knob)
#endif

namespace pnp {

__device__ c32 g_tws[256];

struct SliceArgs {
    float* z;                 // state in slice order, state_stride floats per slice, updated in place
    float* w;
    float* x;                 // written by the last iteration
    const c32* Yh;            // [B] x YH3_SLICE
    const uint32_t* Mh;       // [B] x MH3_SLICE
    const c32* Ys;            // [B][256]   k2 = 128
    const uint32_t* Ms;       // [B][16]
    int first, B, iters;      // slices [first, first + B) of the arrays; iterations of this launch
    int slice_xor;            // experiment knob (PNP_SLICE_XOR): workgroup b takes slice b ^ slice_xor (both below B)
    int flip;                 // workgroup b takes slice B - 1 - b: the launch starts with the slices the previous launch ended with
    int state_stride;         // floats between consecutive slices of z / w (65536 + padding, Slice256::pad)
    int yh_stride;            // complex elements between consecutive slices of Yh (YH3_SLICE + padding)
    float scale, c;
    ProxCoef prox;
    long long* prof;          // phase clock dump of a -DSLICE_PROF build (PNP_SLICE_PROF): [block][2 + 6 per iteration] of wall_clock64()
};

// LDS geometry of the transforms.  Rows of RP = 18 complex (144 B): 16-byte aligned, so a lane reads its row by
// ds_read_b128 (256 B/clk; the ds_read2_b64 hipcc forms from 8-byte reads runs at half that), and conflict-free both
// ways: a 16-lane group's b128 reads start at banks 36 t mod 64 = sixteen distinct 4-bank groups, its b64 column
// writes are 16 consecutive values.
constexpr int RP = 18;
constexpr int REGION = 16 * RP;               // one transform group's exchange region (also holds a row pair in natural order: 256)
constexpr int WREG = 4 * REGION;              // complex elements of a wave's private LDS region: 4 transform groups
constexpr int SL_YS = SL_BUF + REGION;            // operands of the packed column's second half (k2 = 128): Ys (256 complex) + Ms (64 words), see col_phase
constexpr int SL_LDS = SL_YS + 256 + 32;       // transposition buffer (the 8 wave regions alias its start) + W256 table + those
static_assert(SL_WAVES * WREG <= SL_BUF, "wave regions must fit in the buffer they alias");

// wave-synchronous ordering of LDS traffic: a wave's LDS instructions execute in order, so no
// s_barrier is needed between lanes of one wave -- only the compiler has to keep the order
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef SLICE_SYNC_SCHED_BARRIER             // experiment knob: also stop the scheduler here (measured slower)
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// The two waves of a SIMD (w and w + 4) run the same program; left alone, the older one wins the issue arbitration all phase
// long, reaches the barrier 2-3 us early and leaves its partner alone on the SIMD, where one wave cannot hide its own LDS
// latencies.  Trading the priority between register sets keeps them side by side: +2-3 % on a quarter-full chip, +1-1.5 % on a
// full one, for every pattern that changes hands at least once (profiles/variants_r03.log; 0110 = younger, older, older, younger)
__device__ __forceinline__ void prio_set(int wv, int set) {
#if SLICE_PRIOSWAP
    const bool older_high = (SLICE_PRIOMASK >> set) & 1;              // compile-time per set
    if (set == 0 || (((SLICE_PRIOMASK >> set) ^ (SLICE_PRIOMASK >> (set - 1))) & 1)) {      // only where the pattern changes
        asm volatile("" : "+s"(wv));          // compared afresh at every site (s_cmp): a boolean kept across the loop cost a register and a spill
        if ((wv < 4) == older_high) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    }
#endif
}

// Hides a lane-dependent value from loop-invariant code motion: without it hipcc precomputes the
// ~100 LDS / table addresses of all phases before the iteration loop and spills them.
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}
// Global memory goes through buffer instructions: a 128-bit descriptor per array in SGPRs, a wave-uniform
// byte offset in an SGPR (soffset) and ONE 32-bit lane offset register (voffset).  With plain pointers
// hipcc folds "uniform base + uniform row offset + lane offset" into a 64-bit per-lane address per access
// (2 VGPRs each, ~80 in the row phase) and spills.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t bufrsrc;
__device__ __forceinline__ bufrsrc make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// State loads carry sc1 (aux bit 4): they are served by L2, not by the CU's vector L1 -- the loop re-reads
// z / w that THIS workgroup stored one iteration earlier inside the same launch (defensive: the vector L1 is
// not refreshed by stores; costs nothing measurable on 16-byte streaming loads).
__device__ __forceinline__ void ld4(bufrsrc r, int voff, int soff, float (&v)[4]) {
    const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 16);
    v[0] = __uint_as_float(q.x); v[1] = __uint_as_float(q.y); v[2] = __uint_as_float(q.z); v[3] = __uint_as_float(q.w);
}
// Stores fold the wave-uniform offset into the lane offset (soffset = 0), on purpose: a 16-byte buffer store
// reads its data registers over several cycles, and a VALU write to them right behind the store needs wait
// states.  hipcc inserts them only when soffset is NOT an SGPR (the rule of older ISAs); with an SGPR soffset
// it let "buffer_store_dwordx4 v[28:31] ...; v_mov_b32 v28, ..." through, and on gfx950 the store then
// wrote the NEW v28 for the last 4 lanes of each 16 -- seen as sporadic wrong z values.
// The BUILD enforces it: `make` compiles every translation unit to assembly and tools/isa_scan.py fails it when any wide
// buffer store with an SGPR soffset has its data registers overwritten inside the hazard window (csrc/Makefile, `check`).
__device__ __forceinline__ void st4(bufrsrc r, int voff, int soff, const float (&v)[4]) {
    u32x4 q;
    q.x = __float_as_uint(v[0]); q.y = __float_as_uint(v[1]); q.z = __float_as_uint(v[2]); q.w = __float_as_uint(v[3]);
    __builtin_amdgcn_raw_buffer_store_b128(q, r, voff + soff, 0, SLICE_ST_AUX);
}
__device__ __forceinline__ c32 ldc(bufrsrc r, int voff, int soff) {
    const u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return mk<float>(__uint_as_float(q.x), __uint_as_float(q.y));
}
struct SliceBufs {
    bufrsrc z, w, x;          // the slice's [256][256] float arrays
    bufrsrc yh, mh, ys, ms;   // its operand tables
};

// Forces a value to be computed HERE: LLVM otherwise sinks the pure unpack arithmetic of the transposition
// below the following barrier to its first use and keeps (spills) the 64 raw registers it was computed from.
__device__ __forceinline__ void pin(c32& v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); }
// keeps memory operations on their side (limits how many loads the scheduler piles up in registers)
__device__ __forceinline__ void mem_fence_compiler() { asm volatile("" ::: "memory"); }

__device__ __forceinline__ float dpp_lane_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
}
__device__ __forceinline__ c32 dpp_lane_xor1(c32 v) { return mk<float>(dpp_lane_xor1(v.x), dpp_lane_xor1(v.y)); }

// own.x - (partner lane's own.y) and own.y + (partner lane's own.x) as ONE DPP instruction each (hipcc otherwise builds the
// partner's value with two v_mov_dpp and a packed add: three instructions per value)
__device__ __forceinline__ float sub_partner(float a, float b) {          // a - dpp(b)
    float r;
    asm("v_subrev_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(b), "v"(a));
    return r;
}
__device__ __forceinline__ float add_partner(float a, float b) {          // a + dpp(b)
    float r;
    asm("v_add_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(b), "v"(a));
    return r;
}

// |a| + w as one instruction.  (Written in C the compiler shares the |.| with the x store of the final iteration, hoists
// 128 v_and in front of that branch and adds packed: three instructions per register pair instead of two.)
__device__ __forceinline__ float abs_plus(float a, float w) {
    float r;
    asm("v_add_f32_e64 %0, |%1|, %2" : "=v"(r) : "v"(a), "v"(w));
    return r;
}

// a - b as ONE packed instruction (hipcc turned the v = z - w in front of the forward transform into four v_sub_f32 per access)
__device__ __forceinline__ f2 sub2(f2 a, f2 b) {
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// The packed-fp32 transform core (f2, addsub_rot, rot2, tmul_v, tmul_s, dft16_pk) lives in fft16.h: every float kernel uses it.

// 16-lane FFT-256 on a[16] (lane t holds index t + 16 j), exchange through the group's region
template <bool INV>
__device__ __forceinline__ void group_fft256(c32 (&ac)[16], const c32* twl, c32* region, int t) {
    // twl is stored per lane in rows of RP: twl[RP t + k] = W256^(t k): one address register + immediate offsets
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4* tw4 = reinterpret_cast<const f4*>(twl + RP * t);       // (tw[2m], tw[2m + 1]), tw[k] = W256^(t k)
    f2* reg2 = reinterpret_cast<f2*>(region);
    f2 a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = to2(ac[k]);
    dft16_pk<INV>(a);
    mem_fence_compiler();                          // twiddles fetched after the butterflies (not piled up in registers before them)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const f4 q = tw4[m];
        if (m) a[2 * m] = tmul_v<INV>(a[2 * m], k2(q.x, q.y));
        a[2 * m + 1] = tmul_v<INV>(a[2 * m + 1], k2(q.z, q.w));
    }
    mem_fence_compiler();
#pragma unroll
    for (int m = 4; m < 8; ++m) {
        const f4 q = tw4[m];
        a[2 * m] = tmul_v<INV>(a[2 * m], k2(q.x, q.y));
        a[2 * m + 1] = tmul_v<INV>(a[2 * m + 1], k2(q.z, q.w));
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) reg2[k * RP + t] = a[k];
    wave_sync();
    const f4* row4 = reinterpret_cast<const f4*>(reg2 + t * RP);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const f4 q = row4[m];
        a[2 * m] = k2(q.x, q.y);
        a[2 * m + 1] = k2(q.z, q.w);
    }
    wave_sync();
    dft16_pk<INV>(a);
#pragma unroll
    for (int k = 0; k < 16; ++k) ac[k] = from2(a[k]);
}

// ------------------------------------------------------------------------------------------
// rows: one register set = the share of 4 row pairs (8 image rows) this wave owns; lane (g, t) holds elements
// t + 16 j of row pair 32 set + 4 wave + g.  z / w are stored in HBM in the SAME order (sl_state_index, slice_layout.h):
// the lane's q-th 16-byte access of a row pair is (row 2r, row 2r + 1) x (j = 2q, 2q + 1) = the register pairs
// a[2q], a[2q + 1], so the whole z- / w-update runs on the transform's registers.  (Until round 3 the state was in
// natural order and every set went through LDS twice to meet it; that was a third of the kernel's LDS traffic.)
// The z / w values of a set are FETCHED AHEAD (issued before the previous set's forward transform,
// consumed after this set's inverse transform), so HBM latency hides behind the wave's own FFT work.
// ------------------------------------------------------------------------------------------
struct RowLoads {              // [q][(j & 1) * 2 + sel]
    float z[8][4], w[8][4];
};
__device__ __forceinline__ int row_set_offset(int set, int wv) { return (32 * set + 4 * wv) * 2048; }     // bytes; wave-uniform
constexpr int ROW_QSTRIDE = 256;                                  // bytes between a lane's consecutive accesses of a row pair

// voff = 2048 g + 16 t (bytes inside the wave's 8 KiB of a set); the q-th access adds 256 q as an instruction offset
template <int PROX, bool HAS_INV, int Q0, int Q1>
__device__ __forceinline__ void issue_row_loads(const SliceBufs& b, RowLoads& L, int soff, int voff, int qbase = 0) {
#pragma unroll
    for (int q = qbase + Q0; q < qbase + Q1; ++q) {
        const int vo = voff + ROW_QSTRIDE * q, so = soff;
        if (PROX == 3) {                                                   // single-state ADMM_L1: only the w buffer (it carries u)
            ld4(b.w, vo, so, L.w[q]);
        } else if (PROX != 0 || !HAS_INV) {
            ld4(b.z, vo, so, L.z[q]);
            ld4(b.w, vo, so, L.w[q]);
        }
    }
}

// soft(a, c) = a - clamp(a, -c, c): the value of soft_thr (fft16.h) bit for bit (only a zero's sign can differ), as one
// v_med3_f32 per pixel and one packed subtraction per pixel pair instead of compare / select chains
__device__ __forceinline__ f2 clamp2(f2 a, float c) {
    return k2(__builtin_amdgcn_fmed3f(a.x, -c, c), __builtin_amdgcn_fmed3f(a.y, -c, c));
}
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float v) { return k2(v, v); }
// z / w updates of TWO pixels -- the same column of image rows 2r and 2r + 1 (prox_l1_pt / prox_cnc_pt of fft16.h, same
// operation order per pixel)
template <int PROX>
__device__ __forceinline__ void prox_pair(f2 u, f2& z, f2& w, const ProxCoef& pc) {      // u = x + w
    if (PROX == 2) {
        const f2 cz = clamp2(z, pc.ib);                                          // z - soft(z, 1/b)
        const f2 t = fma2(splat(pc.c1), z, fma2(splat(pc.c2), u, splat(pc.c3) * cz));
        z = t - clamp2(t, pc.thr);
    } else {
        z = u - clamp2(u, pc.thr);
    }
    w = u - z;
}

// Pointwise phase of one 16-byte access: the lane's registers a0 = a[2q], a1 = a[2q + 1] (re = image row 2r, im = row 2r + 1)
// with their z / w values; the arithmetic of pointwise4 (fused_pointwise.h).  vs = byte offset of the access in the slice.
// `last` (wave-uniform, run time): the final iteration of a launch stores z and w also in ADMM_L1's single-state form;
// the loop has ONE row-phase body for all its iterations (the kernel is 80+ KB of code and the instruction cache 64 KB;
// a second instance for the final iteration spills 350-490 bytes per lane), at the price of one unused forward
// transform per launch.
template <bool HAS_INV, int PROX, bool HAS_FWD>
__device__ __forceinline__ void pointwise_q(const SliceBufs& b, const ProxCoef& pc, int u_first, bool last,
                                            c32& a0, c32& a1, const float (&z_)[4], const float (&w_)[4], int vs) {
    f2 z[2] = {k2(z_[0], z_[1]), k2(z_[2], z_[3])}, w[2] = {k2(w_[0], w_[1]), k2(w_[2], w_[3])};
    // u = x + w with x = |re|, |im| (the 1/N of the inverse transform is already in the field, col_phase): two v_add_f32 with
    // the |.| source modifier per register pair -- a packed add has no such modifier and would cost two v_and on top
#define SL_X_PLUS(a_, w_) (HAS_INV ? k2(abs_plus((a_).x, (w_).x), abs_plus((a_).y, (w_).y)) : (w_))
#define SL_ST4(buf, v) { const float q_[4] = {v[0].x, v[0].y, v[1].x, v[1].y}; st4(buf, vs, 0, q_); }
    if (PROX == 3) {                                   // ADMM_L1 single-state form: the w buffer carries u = x + w_old
        f2 u[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            if (!u_first) {                            // wave-uniform: a scalar branch (the asm keeps hipcc from turning it into two selects per pair)
                asm volatile("");
                w[jj] = w[jj] - (w[jj] - clamp2(w[jj], pc.thr));
            }
            u[jj] = SL_X_PLUS(jj ? a1 : a0, w[jj]);
            z[jj] = u[jj] - clamp2(u[jj], pc.thr);
            w[jj] = u[jj] - z[jj];
        }
        if (!last) {
            SL_ST4(b.w, u);
        } else {
            SL_ST4(b.z, z);
            SL_ST4(b.w, w);
        }
    }
    if (PROX == 1 || PROX == 2) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) prox_pair<PROX>(SL_X_PLUS(jj ? a1 : a0, w[jj]), z[jj], w[jj], pc);
        SL_ST4(b.z, z);
        SL_ST4(b.w, w);
    }
#undef SL_ST4
#undef SL_X_PLUS
    if (HAS_FWD) {
        const f2 v0 = sub2(z[0], w[0]), v1 = sub2(z[1], w[1]);
        a0 = from2(v0);
        a1 = from2(v1);
    }
}

// x of the last iteration leaves in NATURAL order (it is the caller's result): the set's x = |re|, |im| / N goes through
// the wave's four exchange regions once -- region g holds row pair g of the set as 256 complex (row 2r, row 2r + 1) -- and
// every lane stores 4 consecutive pixels of both rows of each pair.  Once per launch.
__device__ __forceinline__ void store_x_natural(const SliceBufs& b, const c32 (&a)[16], c32* wreg, int set, int wv, int lane) {
    const int g = lane >> 4, t = lane & 15;
    c32* region = wreg + g * REGION;
#pragma unroll
    for (int j = 0; j < 16; ++j) region[t + 16 * j] = mk<float>(fabsf(a[j].x), fabsf(a[j].y));
    wave_sync();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const c32* cell = wreg + i * REGION + 4 * lane;
        const float4 c01 = *reinterpret_cast<const float4*>(cell);
        const float4 c23 = *reinterpret_cast<const float4*>(cell + 2);
        const int vs = 16 * lane + (2 * (32 * set + 4 * wv + i)) * 1024;
        const float xa[4] = {c01.x, c01.z, c23.x, c23.z}, xb[4] = {c01.y, c01.w, c23.y, c23.w};
        st4(b.x, vs, 0, xa);
        st4(b.x, vs + 1024, 0, xb);
    }
    wave_sync();
}

// all four register sets of a wave; loads of set s + 1 are in flight during the transforms around them
#ifndef SLICE_L1_PF
#define SLICE_L1_PF 8       // ADMM_L1's single-state form (PROX 3): ALL eight accesses of the next set are in flight across this set's forward and the
#endif                      // next set's inverse transform -- the overlap the form's 8 N fewer bytes leave room for (experiment knob: 4, 2 = less of it)
template <int PROX> constexpr int row_pf() { return (PROX == 3) ? SLICE_L1_PF : SLICE_PF; }
// the first accesses of set 0, issued by the caller ahead of the phase (SLICE_EARLY: between the two passes of T2)
template <int PROX, bool HAS_INV>
__device__ __forceinline__ void row_phase_prefetch(const SliceBufs& b, RowLoads& L, int wv, int lane) {
    issue_row_loads<PROX, HAS_INV, 0, row_pf<PROX>()>(b, L, row_set_offset(0, wv), 2048 * (lane >> 4) + 16 * (lane & 15));
}
template <bool HAS_INV, int PROX, bool HAS_FWD, bool PRELOADED = false>
__device__ __forceinline__ void row_phase(const SliceBufs& b, const ProxCoef& pc, int u_first, bool last, c32 (&F)[SL_SETS][16],
                                          c32* wreg, const c32* twl, int wv, int lane, RowLoads* pre = nullptr) {
    const int g = lane >> 4, t = lane & 15;
    c32* region = wreg + g * REGION;
    const int voff = 2048 * g + 16 * t;
    // PF accesses of the next set are fetched ahead across the transforms; the rest when the set's pointwise phase starts
    constexpr int PF = row_pf<PROX>();
    RowLoads L;
    if (PRELOADED) L = *pre;
    else issue_row_loads<PROX, HAS_INV, 0, PF>(b, L, row_set_offset(0, wv), voff);
#pragma unroll
    for (int set = 0; set < SL_SETS; ++set) {
        c32 (&a)[16] = F[set];
        prio_set(wv, set);
        const int soff = row_set_offset(set, wv);
        if (HAS_INV) {
            group_fft256<true>(a, twl, region, t);
        }
        if (HAS_INV && last) store_x_natural(b, a, wreg, set, wv, lane);
        const int vs = voff + soff;
        // rolling fetch: access q + PF goes out when access q is consumed (PF accesses = 8 PF registers in flight; all 8 at
        // once, on top of the 128 data registers, made hipcc spill)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q + PF < 8) issue_row_loads<PROX, HAS_INV, 0, 1>(b, L, soff, voff, q + PF);
            pointwise_q<HAS_INV, PROX, HAS_FWD>(b, pc, u_first, last, a[2 * q], a[2 * q + 1], L.z[q], L.w[q], vs + ROW_QSTRIDE * q);
        }
        if (set + 1 < SL_SETS) issue_row_loads<PROX, HAS_INV, 0, PF>(b, L, row_set_offset(set + 1, wv), voff);
        if (HAS_FWD) {
            group_fft256<false>(a, twl, region, t);
        }
    }
}

// ------------------------------------------------------------------------------------------
// transpositions (workgroup-wide, two passes each); slots: slice_layout.h
// ------------------------------------------------------------------------------------------
// The 8 registers of a row-form set that cross in pass P, with their slots (sl_pass / sl_slot of
// slice_layout.h, k = t + 16 j).  Register indices must be literals (register arrays), so the lists
// are spelled out: X(ja, jb, slot) handles register ja -- or, in lane 0 of a group, register jb:
// k = 128 and k = 192 sit in lane 0 of j = 8 / 12 and belong to the other pass than the rest of their register.
#define SL_PASS0_REGS(X, t)                                                                       \
    X(0, 0, (t)) X(1, 1, (t) + 16) X(2, 2, (t) + 32) X(3, 3, (t) + 48)                              \
    X(12, 8, ((t) ? SL_M + 64 - (t) : SL_M))                                                      \
    X(13, 13, SL_M + 48 - (t)) X(14, 14, SL_M + 32 - (t)) X(15, 15, SL_M + 16 - (t))
#define SL_PASS1_REGS(X, t)                                                                       \
    X(4, 4, (t)) X(5, 5, (t) + 16) X(6, 6, (t) + 32) X(7, 7, (t) + 48)                              \
    X(8, 12, ((t) ? SL_M + 64 - (t) : SL_M))                                                      \
    X(9, 9, SL_M + 48 - (t)) X(10, 10, SL_M + 32 - (t)) X(11, 11, SL_M + 16 - (t))

// The MIRROR half of a buffer row (slots SL_M ..) holds its values with re and im SWAPPED.  Both lanes of a pair then run
// the same two instructions in both transpositions (no selects by lane parity): with own = the lane's value as stored and
// (.)' = the partner lane's, T1 needs ((own.x + own.y'), (own.y - own.x')) / 2 and T2 writes (own.x - own.y', own.y + own.x').
// The swap itself is free: a swapped 8-byte access is a ds_write2_b32 / ds_read2_b32 with its two dword offsets exchanged.
__device__ __forceinline__ void lds_put(c32* p, c32 v, bool swapped) {
    float* f = reinterpret_cast<float*>(p);
    if (swapped) { f[0] = v.y; f[1] = v.x; } else { *p = v; }
}
__device__ __forceinline__ c32 lds_get(const c32* p, bool swapped) {
    const float* f = reinterpret_cast<const float*>(p);
    return swapped ? mk<float>(f[1], f[0]) : *p;
}
// row-form registers of one set -> buffer row `rp`
template <int P>
__device__ __forceinline__ void t_store_rows(const c32 (&F)[16], c32* rp, int t) {
#define SL_T1_STORE(ja, jb, slot) { lds_put(rp + (slot), t ? F[ja] : F[jb], (ja) >= 8); }
    if (P == 0) { SL_PASS0_REGS(SL_T1_STORE, t) } else { SL_PASS1_REGS(SL_T1_STORE, t) }
#undef SL_T1_STORE
}
// buffer row `rp` -> row-form registers of one set (lane 0 of a group is fixed up after pass 1)
template <int P>
__device__ __forceinline__ void t_load_rows(c32 (&F)[16], const c32* rp, int t) {
#define SL_T2_LOAD(ja, jb, slot) { F[ja] = lds_get(rp + (slot), (ja) >= 8); }
    if (P == 0) { SL_PASS0_REGS(SL_T2_LOAD, t) } else { SL_PASS1_REGS(SL_T2_LOAD, t) }
#undef SL_T2_LOAD
    if (P == 1) {         // lane 0 received k = 128 in register 12 (pass 0) and k = 192 in register 8 (pass 1): swap them
        const c32 a0 = F[8];
        F[8] = t ? a0 : F[12];
        F[12] = t ? F[12] : a0;
    }
}

// T1 read side.  (d, m) = (C_r[c], C_r[-c]) of a row pair give the row transforms of its two REAL image rows at column c:
//   even lane: unpack_a(d, m) = (d + conj m) / 2,   odd lane: unpack_b(d, m) = (d - conj m) / (2 i)     (fft16.h).
// The even lane of a pair reads d, the odd lane m with its halves swapped (see above) -- ONE 8-byte LDS read per lane and
// value -- and the partner's halves come by DPP inside the add / subtract:
//   2 x value = ( own.x + partner's own.y,  own.y - partner's own.x )     -- the sums and differences of unpack_a / unpack_b, term for term;
//   the column phase works on the doubled field and takes the 1/2 back in its blend coefficient (exactly: a power of two)
// The 16 reads of a column go out in two groups of 8 (until round 3 every read was waited for on its own: 64 serial LDS
// round trips per wave and iteration).
// One column-form set's 16 values from the buffer.  MAYBE_PACKED: this wave owns the packed column c = 0 (wave 0, first
// set of pass 0, lanes 0..15 -- `packed` says which lanes); every other (wave, set) runs the plain two-instruction unpack.
// Buffer rows 64.. (j >= 8) lie beyond the 16-bit offset of a DS instruction: a second base keeps the offsets immediates.
template <bool MAYBE_PACKED>
__device__ __forceinline__ void t1_read_col(c32 (&Gs)[16], const c32* buf, int off, int odd, bool packed) {
    int off_hi = off + 64 * SL_P;
    asm volatile("" : "+v"(off_hi));                // an index, not a pointer: the address space stays visible to the compiler
    const c32 *col = buf + off, *col_hi = buf + off_hi;
#pragma unroll
    for (int jb = 0; jb < 16; jb += 8) {
        c32 own[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) own[k] = jb ? col_hi[8 * k * SL_P] : col[8 * k * SL_P];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            // TWICE the unpacked value: the 1/2 is folded into the blend coefficients (blend_scaled, exact)
            c32 v = mk<float>(add_partner(own[k].x, own[k].y), sub_partner(own[k].y, own[k].x));
            if (MAYBE_PACKED) {                     // the packed column c = 0 takes the raw values: (C[0], C[128]) -> re / im parts
                const float oy = dpp_lane_xor1(own[k].y);
                const c32 raw = mk<float>(odd ? oy : own[k].x, odd ? own[k].x : oy);
                v = packed ? raw : v;
            }
            Gs[jb + k] = v;
            pin(Gs[jb + k]);                        // unpack as the values arrive: raw values must not pile up across the barrier
        }
    }
}
template <int P>
__device__ __forceinline__ void t1_pass(const c32 (&F)[SL_SETS][16], c32 (&G)[SL_SETS][16], c32* buf, int wv, int lane) {
    const int g = lane >> 4, t = lane & 15;
#pragma unroll
    for (int set = 0; set < SL_SETS; ++set) t_store_rows<P>(F[set], buf + (32 * set + 4 * wv + g) * SL_P, t);
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {                       // the two column-form sets of this pass: 2P and 2P + 1
        const int cc = 32 * h + 4 * wv + g, odd = t & 1;
        const int off = (t >> 1) * SL_P + cc + (odd ? SL_M : 0);
        if (P == 0 && h == 0 && wv == 0) t1_read_col<true>(G[2 * P + h], buf, off, odd, cc == 0);      // wv is wave-uniform: a scalar branch
        else t1_read_col<false>(G[2 * P + h], buf, off, odd, false);
    }
    __syncthreads();
}

template <bool MAYBE_PACKED>
__device__ __forceinline__ void t2_write_col(const c32 (&Gs)[16], c32* buf, int off, int odd, bool packed) {
    int off_hi = off + 64 * SL_P;
    asm volatile("" : "+v"(off_hi));
    c32 *col = buf + off, *col_hi = buf + off_hi;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        // the lane pair (even, odd) holds (ue, uo) = column values of image rows 2r, 2r + 1; the even lane writes
        // repack_p(ue, uo) = ue + i uo to the direct slot, the odd lane repack_q(ue, uo) = conj ue + i conj uo to the mirror
        // slot -- with its halves swapped, which makes both lanes' values ( own.x - partner's own.y,  own.y + partner's own.x )
        const c32 own = Gs[j];
        c32 v = mk<float>(sub_partner(own.x, own.y), add_partner(own.y, own.x));
        if (MAYBE_PACKED) {                         // packed column: (ue.x, uo.x) direct, (ue.y, uo.y) mirror (stored swapped)
            const c32 other = dpp_lane_xor1(own);
            const c32 raw = mk<float>(odd ? own.y : own.x, odd ? other.y : other.x);
            v = packed ? raw : v;
        }
        if (j < 8) col[8 * j * SL_P] = v; else col_hi[8 * (j - 8) * SL_P] = v;
    }
}
template <int P>
__device__ __forceinline__ void t2_pass(const c32 (&G)[SL_SETS][16], c32 (&F)[SL_SETS][16], c32* buf, int wv, int lane) {
    const int g = lane >> 4, t = lane & 15;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int cc = 32 * h + 4 * wv + g, odd = t & 1;
        const int off = (t >> 1) * SL_P + cc + (odd ? SL_M : 0);
        if (P == 0 && h == 0 && wv == 0) t2_write_col<true>(G[2 * P + h], buf, off, odd, cc == 0);     // wv is wave-uniform: a scalar branch
        else t2_write_col<false>(G[2 * P + h], buf, off, odd, false);
    }
    __syncthreads();
#pragma unroll
    for (int set = 0; set < SL_SETS; ++set) t_load_rows<P>(F[set], buf + (32 * set + 4 * wv + g) * SL_P, t);
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// columns: one register set = the share of 4 columns this wave owns; operands fetched one set ahead
// ------------------------------------------------------------------------------------------
struct ColLoads {
    c32 yh[16];
    uint32_t code;
};
__device__ __forceinline__ void issue_col_loads(const SliceBufs& b, ColLoads& Y, int set, int wv, int lane) {
    const int ybase = (set * 8 + wv) * 8192;                       // the wave's 8 KiB block of the set: yh3_index(.., set, j = 0, wv, lane = 0) in bytes
    Y.code = __builtin_amdgcn_raw_buffer_load_b32(b.mh, 4 * lane, (set * 8 + wv) * 64 * 4, 0);
#pragma unroll
    for (int jp = 0; jp < 8; ++jp) {                               // (j = 2 jp, 2 jp + 1) per 16-byte access; offsets 0 .. 3072 are instruction immediates
        const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(b.yh, 16 * lane + 1024 * (jp & 3), ybase + 4096 * (jp >> 2), SLICE_YH_AUX);
        Y.yh[2 * jp] = mk<float>(__uint_as_float(q.x), __uint_as_float(q.y));
        Y.yh[2 * jp + 1] = mk<float>(__uint_as_float(q.z), __uint_as_float(q.w));
    }
}

// blend_scaled (fft16.h) over a lane's 16 values, the same arithmetic in fewer instructions: the code word holds 2 bits per j at bit
// 2 j = byte j / 4, bits 2 (j % 4), so four masked copies put every code into a byte of its own and v_cvt_f32_ubyteN converts it
// straight from there (hipcc: a v_bfe_u32 and a v_cvt per value), and the coefficients A = fma(-chs, code, os) of a pair (j, j + 1)
// come out of one packed fma, each half broadcast by the blend's op_sel.  63 instructions per set instead of 78.
__device__ __forceinline__ float code_byte(unsigned m, int byte) {
    float r;
    if (byte == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(r) : "v"(m));
    else if (byte == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(r) : "v"(m));
    else if (byte == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(r) : "v"(m));
    else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(r) : "v"(m));
    return r;
}
__device__ __forceinline__ void blend_set(c32 (&a)[16], const c32 (&yh)[16], unsigned code, float cs, float chs, float os) {
    const unsigned m[4] = {code & 0x03030303u, (code >> 2) & 0x03030303u, (code >> 4) & 0x03030303u, (code >> 6) & 0x03030303u};
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        const f2 cf = k2(code_byte(m[j & 3], j >> 2), code_byte(m[(j & 3) + 1], j >> 2));
        const f2 A2 = fma2(splat(-chs), cf, splat(os));
        a[j] = from2(fma2(A2.xx, to2(a[j]), splat(cs) * to2(yh[j])));
        a[j + 1] = from2(fma2(A2.yy, to2(a[j + 1]), splat(cs) * to2(yh[j + 1])));
    }
}

__device__ __forceinline__ void col_phase(const SliceBufs& b, float cdc, float scale, c32 (&G)[SL_SETS][16], c32* wreg, c32* ysl, const c32* twl, int wv, int lane) {
    const int g = lane >> 4, t = lane & 15;
    c32* region = wreg + g * REGION;
    // the blend also applies the inverse transforms' 1/N (scale, a power of two): every coefficient of blend_scaled carries it,
    // the blended field is exactly scale x blend_one's, and the inverse transforms deliver x without a multiplication per pixel
    const float cs = cdc * scale, chs = 0.5f * cdc * scale;
    ColLoads Y;
    issue_col_loads(b, Y, 0, wv, lane);
    if (wv == 0) {
        // wave 0 also owns the packed column's second half.  Its operands (2 KiB + 16 words) go straight to LDS
        // (`buffer_load ... lds`, no registers) while the first transform runs: fetched inside the branch below they
        // cost this wave a full memory latency, and the other seven waited for it at the barrier (2.5 us per iteration)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(b.ys, ysl, 16, 16 * lane, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(b.ys, ysl + 128, 16, 16 * lane, 1024, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(b.ms, ysl + 256, 4, 4 * lane, 0, 0, 0);       // lanes 16.. are out of range: zeros
    }
#pragma unroll
    for (int set = 0; set < SL_SETS; ++set) {
        c32 (&a)[16] = G[set];
        prio_set(wv, set);
        group_fft256<false>(a, twl, region, t);                   // a[j] = spectrum at k1 = t + 16 j, k2 = c
        if (set == 0 && wv == 0 && g == 0) {
            // packed column: a = A + i B, A / B = spectra of the real columns k2 = 0 / 128; split with the mirror k1 -> -k1.
            // The region holds the column plus a wrap-around copy of its first 16 values, so that the mirror of
            // k1 = t + 16 j is region[256 - k1] = (region + 16 - t)[16 (15 - j)] for every (t, j): one address, immediates.
#pragma unroll
            for (int j = 0; j < 16; ++j) region[t + 16 * j] = a[j];
            region[256 + t] = a[0];
            wave_sync();
            const uint32_t code_b = reinterpret_cast<const uint32_t*>(ysl + 256)[t];
            const c32* mirror = region + (16 - t);
            const unsigned ma[4] = {Y.code & 0x03030303u, (Y.code >> 2) & 0x03030303u, (Y.code >> 4) & 0x03030303u, (Y.code >> 6) & 0x03030303u};
            const unsigned mb[4] = {code_b & 0x03030303u, (code_b >> 2) & 0x03030303u, (code_b >> 4) & 0x03030303u, (code_b >> 6) & 0x03030303u};
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const c32 gm = mirror[16 * (15 - j)];
                const c32 A = blend_scaled_f(unpack_a(a[j], gm), Y.yh[j], code_byte(ma[j & 3], j >> 2), cs, chs, scale);
                const c32 Bv = blend_scaled_f(unpack_b(a[j], gm), ysl[t + 16 * j], code_byte(mb[j & 3], j >> 2), cs, chs, scale);
                a[j] = repack_p(A, Bv);
            }
            wave_sync();
        } else {
            blend_set(a, Y.yh, Y.code, cs, 0.5f * chs, 0.5f * scale);        // doubled field: half the coefficients
        }
        if (set + 1 < SL_SETS) issue_col_loads(b, Y, set + 1, wv, lane);
        group_fft256<true>(a, twl, region, t);                    // column c of the blended field, unnormalised
    }
}

// ------------------------------------------------------------------------------------------
// PROX: 1 L1 (z and w), 2 CNC, 3 L1 single-state (see fused_pointwise.h)
// ------------------------------------------------------------------------------------------
template <int PROX>
__global__ __launch_bounds__(512) void k_slice(SliceArgs p) {
    __shared__ __attribute__((aligned(16))) c32 lds[SL_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave index as a scalar: bases below stay in SGPRs
    c32* twl = lds + SL_BUF;
    if (tid < 256) twl[RP * (tid >> 4) + (tid & 15)] = g_tws[((tid >> 4) * (tid & 15)) & 255];     // [t][k] = W256^(t k), rows of RP
    c32* wreg = lds + wv * WREG;
    __syncthreads();
    for (int sb = blockIdx.x; sb < p.B; sb += gridDim.x) {
        const int sx = ((sb ^ p.slice_xor) < p.B) ? (sb ^ p.slice_xor) : sb;
        const int slice = p.first + (p.flip ? p.B - 1 - sx : sx);
        const size_t so = (size_t)slice * p.state_stride;
        SliceBufs b;
        b.z = make_rsrc(p.z + so, 65536 * 4); b.w = make_rsrc(p.w + so, 65536 * 4); b.x = make_rsrc(p.x + (size_t)slice * 65536, 65536 * 4);
        b.yh = make_rsrc(p.Yh + (size_t)slice * p.yh_stride, YH3_SLICE * 8); b.mh = make_rsrc(p.Mh + (size_t)slice * MH3_SLICE, MH3_SLICE * 4);
        b.ys = make_rsrc(p.Ys + (size_t)slice * 256, 256 * 8); b.ms = make_rsrc(p.Ms + (size_t)slice * 16, 16 * 4);
        c32 F[SL_SETS][16];
        // phase clocks exist only in a -DSLICE_PROF build (profiles/variants.sh): in the product they cost two registers
        // and a dozen branches of a kernel that has neither to spare
#ifdef SLICE_PROF
        long long* prof = (p.prof && tid == 0 && sb == (int)blockIdx.x) ? p.prof + (size_t)slice * (2 + 6 * p.iters) : nullptr;
#define SL_STAMP() if (prof) *prof++ = wall_clock64()
#else
#define SL_STAMP()
#endif
        SL_STAMP();
        row_phase<false, 0, true>(b, p.prox, 1, false, F, wreg, twl, wv, opaque(lane));
        SL_STAMP();
        for (int it = 0; it < p.iters; ++it) {
            c32 G[SL_SETS][16];
            __syncthreads();                      // every wave is done with its private region: the buffer aliases them
            SL_STAMP();                           // wave 0's wait for the slowest wave of the row phase ends here
            t1_pass<0>(F, G, lds, wv, opaque(lane));
            t1_pass<1>(F, G, lds, wv, opaque(lane));
#ifdef SLICE_PROF_CLOCK                      // diagnostic build: this slot carries the SHADER clock (s_memtime) instead of the 100 MHz wall clock
            if (prof) *prof++ = (long long)__builtin_readcyclecounter();
#else
            SL_STAMP();
#endif
            col_phase(b, p.c, p.scale, G, wreg, lds + SL_YS, twl, wv, opaque(lane));
            SL_STAMP();
            __syncthreads();
            SL_STAMP();
            t2_pass<0>(G, F, lds, wv, opaque(lane));
#if SLICE_EARLY
            RowLoads L0;
            row_phase_prefetch<PROX, true>(b, L0, wv, opaque(lane));
#endif
            t2_pass<1>(G, F, lds, wv, opaque(lane));
            SL_STAMP();
            const int u_first = (it == 0);
#if SLICE_EARLY
            row_phase<true, PROX, true, true>(b, p.prox, u_first, it + 1 == p.iters, F, wreg, twl, wv, opaque(lane), &L0);
#else
            row_phase<true, PROX, true>(b, p.prox, u_first, it + 1 == p.iters, F, wreg, twl, wv, opaque(lane));
#endif
            SL_STAMP();
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// operand tables (once per uploaded problem): one block per (tile of 16 columns k2, slice); tile 8 = the column k2 = 128.
// The tile of y (and of the mask) and its mirror image (rows -k1, columns -k2) go through LDS, so that global memory is
// read in 128-byte row segments and every table block is written in its own contiguous order.  (Until round 3 a block held
// ONE column and its 256 threads read y with a 2 KiB stride: 2.3 GB fetched for 0.4 GB of input at 512 slices, 350 us.)
// Arithmetic = hermitian_entry_t (fused_layout.h): unsampled entries are selected away, never multiplied.
// ------------------------------------------------------------------------------------------
constexpr int SP_P = 17;                       // tile pitch (columns + 1)
__global__ __launch_bounds__(256) void k_sprepare(const c32* y, const uint8_t* mask_bank, const int32_t* mask_id,
                                                  c32* Yh, uint32_t* Mh, c32* Ys, uint32_t* Ms, int yh_stride) {
    __shared__ c32 yd[256 * SP_P], ym[256 * SP_P];           // direct tile [row][c], mirror tile [row][c] = y[row][-(16 m + c)]
    __shared__ uint8_t md[256 * SP_P], mm[256 * SP_P];
    const int tid = threadIdx.x, m = blockIdx.x, slice = blockIdx.y;
    const int mid = mask_id ? mask_id[slice] : 0;
    const c32* ys = y + (size_t)slice * 65536;
    const uint8_t* ms = mask_bank + (size_t)mid * 65536;
    const int ncol = (m == 8) ? 1 : 16;                        // tile 8: k2 = 128 alone (its own mirror column)
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int idx = tid + 256 * i, r = idx >> 4, c = idx & 15;
        if (c < ncol) {
            const int k2 = 16 * m + c, k2m = (256 - k2) & 255;
            yd[r * SP_P + c] = ys[r * 256 + k2];
            md[r * SP_P + c] = ms[r * 256 + k2];
            ym[r * SP_P + c] = ys[r * 256 + k2m];
            mm[r * SP_P + c] = ms[r * 256 + k2m];
        }
    }
    __syncthreads();
    auto entry = [&](int k1, int c, c32& yh) -> int {
        const int r2 = (256 - k1) & 255;
        const int m1 = md[k1 * SP_P + c] != 0, m2 = mm[r2 * SP_P + c] != 0;
        const c32 y1 = yd[k1 * SP_P + c], y2 = ym[r2 * SP_P + c];
        yh = mk<float>(0.5f * ((m1 ? y1.x : 0.f) + (m2 ? y2.x : 0.f)), 0.5f * ((m1 ? y1.y : 0.f) - (m2 ? y2.y : 0.f)));
        return m1 + m2;
    };
    if (m == 8) {                                              // k2 = 128: Ys [256], Ms [16] (lane t, bits j)
        c32 yh;
        const int code = entry(tid, 0, yh);
        Ys[(size_t)slice * 256 + tid] = yh;
        __shared__ int codes[256];
        codes[tid] = code;
        __syncthreads();
        if (tid < 16) {
            uint32_t v = 0;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) v |= (uint32_t)codes[tid + 16 * jj] << (2 * jj);
            Ms[slice * 16 + tid] = v;
        }
        return;
    }
    // the tile's 16 columns = four (set, wave) blocks of the table, 1024 entries each, written in storage order
    const int set = m >> 1;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int o = tid + 256 * i, blk = o >> 10, e = o & 1023;           // block blk: columns 4 blk .. 4 blk + 3 of the tile
        const int jp = e >> 7, lane = (e >> 1) & 63, j = 2 * jp + (e & 1);
        const int g = lane >> 4, t = lane & 15, wv = 4 * (m & 1) + blk;
        c32 yh;
        (void)entry(t + 16 * j, 4 * blk + g, yh);
        Yh[(size_t)slice * yh_stride + yh3_index(0, set, j, wv, lane)] = yh;
    }
    {
        const int blk = tid >> 6, lane = tid & 63, g = lane >> 4, t = lane & 15, wv = 4 * (m & 1) + blk;
        uint32_t v = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            c32 yh;
            v |= (uint32_t)entry(t + 16 * j, 4 * blk + g, yh) << (2 * j);
        }
        Mh[mh3_index(slice, set, wv, lane)] = v;
    }
}

// ------------------------------------------------------------------------------------------
// natural order <-> slice order of the state arrays (sl_state_index), in place or between the caller-visible arrays and the
// kernel's padded ones (Slice256::pad): one block per (4 row pairs, slice), 128
// threads x 4 x 16 bytes per array.  Runs when a context's loops switch kernel families or the caller reads / writes the state.
// ------------------------------------------------------------------------------------------
template <bool TO_SLICE>
__global__ __launch_bounds__(128) void k_state_order(const float* sz, const float* sw, float* dz, float* dw, int src_stride, int dst_stride) {
    __shared__ float tile[2][2048];                       // one chunk = 4 row pairs = 8 image rows of the slice, both arrays
    const size_t sbase = (size_t)blockIdx.y * src_stride + (size_t)blockIdx.x * 2048, dbase = (size_t)blockIdx.y * dst_stride + (size_t)blockIdx.x * 2048;
    const float4* pz = reinterpret_cast<const float4*>(sz + sbase);
    const float4* pw = reinterpret_cast<const float4*>(sw + sbase);
    float4* qz = reinterpret_cast<float4*>(dz + dbase);
    float4* qw = reinterpret_cast<float4*>(dw + dbase);
    float4 vz[4], vw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { vz[u] = pz[threadIdx.x + 128 * u]; vw[u] = pw[threadIdx.x + 128 * u]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = threadIdx.x + 128 * u, row = i >> 6, n0 = 4 * (i & 63);        // natural: float4 i = row (0..7) x pixels n0..n0+3
        const float az[4] = {vz[u].x, vz[u].y, vz[u].z, vz[u].w}, aw[4] = {vw[u].x, vw[u].y, vw[u].z, vw[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int at = TO_SLICE ? (int)sl_state_index(row, n0 + k) : 4 * i + k;
            tile[0][at] = az[k];
            tile[1][at] = aw[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = threadIdx.x + 128 * u, row = i >> 6, n0 = 4 * (i & 63);
        float oz[4], ow[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int at = TO_SLICE ? 4 * i + k : (int)sl_state_index(row, n0 + k);
            oz[k] = tile[0][at];
            ow[k] = tile[1][at];
        }
        qz[i] = make_float4(oz[0], oz[1], oz[2], oz[3]);
        qw[i] = make_float4(ow[0], ow[1], ow[2], ow[3]);
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct Slice256 {
    int Bmax = 0, cus = 0;
    int flip = 0;                            // direction of the next multi-round launch
    // Slice stride of the kernel's own arrays.  pad > 0: the state in slice order lives in zs / ws with 65536 + pad floats per slice
    // (the caller-visible z / w keep their contiguous natural layout); yh_pad likewise for the Hermitian table.  (PNP_SLICE_PAD_KB)
    int pad = 0, yh_pad = 0;
    float* zs = nullptr;
    float* ws = nullptr;
    c32* Yh = nullptr;
    uint32_t* Mh = nullptr;
    c32* Ys = nullptr;
    uint32_t* Ms = nullptr;
    static constexpr int MAXQ = 4;
    hipStream_t side[MAXQ - 1] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr;
    hipEvent_t ev_join[MAXQ - 1] = {nullptr, nullptr, nullptr};
};

int slice256_cus(const Slice256* f) { return f && f->cus > 0 ? f->cus : 256; }

hipError_t slice256_state_order(Slice256* f, hipStream_t s, float* z, float* w, int B, bool to_slice) {
    if (B <= 0) return hipSuccess;
    if (f && f->pad > 0) {                                 // between the natural arrays and the padded slice-order arrays
        if (to_slice) hipLaunchKernelGGL(k_state_order<true>, dim3(32, B), dim3(128), 0, s, z, w, f->zs, f->ws, 65536, 65536 + f->pad);
        else          hipLaunchKernelGGL(k_state_order<false>, dim3(32, B), dim3(128), 0, s, f->zs, f->ws, z, w, 65536 + f->pad, 65536);
    } else {                                               // in place
        if (to_slice) hipLaunchKernelGGL(k_state_order<true>, dim3(32, B), dim3(128), 0, s, z, w, z, w, 65536, 65536);
        else          hipLaunchKernelGGL(k_state_order<false>, dim3(32, B), dim3(128), 0, s, z, w, z, w, 65536, 65536);
    }
    return hipGetLastError();
}

void slice256_destroy(Slice256* f) {
    if (!f) return;
    if (f->zs) (void)hipFree(f->zs);
    if (f->ws) (void)hipFree(f->ws);
    if (f->Yh) (void)hipFree(f->Yh);
    if (f->Mh) (void)hipFree(f->Mh);
    if (f->Ys) (void)hipFree(f->Ys);
    if (f->Ms) (void)hipFree(f->Ms);
    for (int q = 0; q < Slice256::MAXQ - 1; ++q) {
        if (f->side[q]) (void)hipStreamDestroy(f->side[q]);
        if (f->ev_join[q]) (void)hipEventDestroy(f->ev_join[q]);
    }
    if (f->ev_fork) (void)hipEventDestroy(f->ev_fork);
    delete f;
}

Slice256* slice256_create(int Bmax, int pad_kb, int yh_pad_kb, hipError_t* err) {
    Slice256* f = new Slice256();
    f->Bmax = Bmax;
    // 4 KiB of padding behind every slice of the state and of the table: with 256 KiB strides the streams of all resident
    // workgroups sit on the same address bits above bit 17 at the same time; +2.4 % at the driver's 20 steps (9740 -> 9977 it/s,
    // five alternating runs, variants_r03.log v28), +-0 at 100.  0 = round 3's in-place, unpadded form.
    // (PNP_SLICE_PAD_KB / PNP_SLICE_YH_PAD_KB, read and range-checked at pnp_ctx_create)
    f->pad = (pad_kb > 0 ? pad_kb : 0) * 256;                 // floats
    f->yh_pad = (yh_pad_kb > 0 ? yh_pad_kb : 0) * 128;        // complex elements
    hipError_t e = hipMalloc((void**)&f->Yh, (size_t)Bmax * (YH3_SLICE + f->yh_pad) * sizeof(c32));
    if (e == hipSuccess && f->pad > 0) e = hipMalloc((void**)&f->zs, (size_t)Bmax * (65536 + f->pad) * sizeof(float));
    if (e == hipSuccess && f->pad > 0) e = hipMalloc((void**)&f->ws, (size_t)Bmax * (65536 + f->pad) * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&f->Mh, (size_t)Bmax * MH3_SLICE * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&f->Ys, (size_t)Bmax * 256 * sizeof(c32));
    if (e == hipSuccess) e = hipMalloc((void**)&f->Ms, (size_t)Bmax * 16 * sizeof(uint32_t));
    if (e == hipSuccess) {
        static thread_local c32 h[256];
        for (int m = 0; m < 256; ++m) {
            const double a = -2.0 * M_PI * (double)m / 256.0;
            h[m] = mk<float>((float)cos(a), (float)sin(a));
        }
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_tws), h, sizeof(h));
    }
    if (e == hipSuccess) {
        int dev = 0;
        hipDeviceProp_t prop;
        e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
        if (e == hipSuccess) f->cus = prop.multiProcessorCount;
    }
    if (e != hipSuccess) {
        slice256_destroy(f);
        *err = e;
        return nullptr;
    }
    *err = hipSuccess;
    return f;
}

hipError_t slice256_prepare(Slice256* f, hipStream_t s, const float2* y, const uint8_t* mask_bank, const int32_t* mask_id, int B) {
    if (B > f->Bmax) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_sprepare, dim3(9, B), dim3(256), 0, s, reinterpret_cast<const c32*>(y), mask_bank, mask_id,
                       f->Yh, f->Mh, f->Ys, f->Ms, (int)YH3_SLICE + f->yh_pad);
    return hipGetLastError();
}

static hipError_t launch_slice(hipStream_t s, const SliceArgs& a, int prox) {
    // one workgroup per slice; a workgroup fills a compute unit (512 threads x 256 VGPRs, 134 KiB of LDS)
    const dim3 grid(a.B);
    if (prox == 2)      hipLaunchKernelGGL(k_slice<2>, grid, dim3(512), 0, s, a);
    else if (prox == 1) hipLaunchKernelGGL(k_slice<1>, grid, dim3(512), 0, s, a);
    else                hipLaunchKernelGGL(k_slice<3>, grid, dim3(512), 0, s, a);
    return hipGetLastError();
}

// Experiment knobs (FusedSchedule::slice_queues / slice_segment; default: ONE launch): a run can be cut into parts of the
// batch on several HIP queues times consecutive launches of a part of the iterations.  Workgroups differ in speed
// by +-10 % (up to +25 %) and with two workgroups per compute unit the slowest unit sets the time of a single
// launch (makespan 8.0 ms against 7.0 ms of balanced work at 50 iterations); smaller units were meant to even this
// out but measured no better (6.3 k it/s for 2 queues x 16-iteration segments against 6.5 k for one launch).
// A run split into calls is bit-identical to one call (tests), so such cuts never change results.
hipError_t slice256_run(Slice256* f, hipStream_t s, float* z, float* w, float* x, int B, int iters, bool cnc, float dc_c,
                        ProxParams pp, const FusedSchedule& sch) {
    if (iters <= 0) return hipSuccess;
    SliceArgs a;
    a.z = f->pad > 0 ? f->zs : z; a.w = f->pad > 0 ? f->ws : w; a.x = x; a.Yh = f->Yh;
    a.state_stride = 65536 + f->pad; a.yh_stride = (int)YH3_SLICE + f->yh_pad; a.Mh = f->Mh; a.Ys = f->Ys; a.Ms = f->Ms;
    a.first = 0; a.B = B; a.iters = iters; a.scale = 1.0f / 65536.0f; a.c = dc_c;
    a.prox.thr = pp.thr; a.prox.c1 = pp.c1; a.prox.c2 = pp.c2; a.prox.c3 = pp.c3; a.prox.ib = pp.ib;
    a.prof = nullptr;
    a.slice_xor = 0;
    a.flip = 0;
    a.slice_xor = sch.slice_xor;
    if (a.slice_xor < 0 || (B & (B - 1)) != 0 || a.slice_xor >= B) a.slice_xor = 0;      // a permutation only for power-of-two batches
    const int prox = cnc ? 2 : (sch.l1_two_state ? 1 : 3);
    int queues = sch.slice_queues, seg_len = sch.slice_segment;                                // measured: 1 launch is best
    if (queues < 1) queues = 1;
    if (queues > Slice256::MAXQ) queues = Slice256::MAXQ;
    if (B < 64 * queues) queues = 1;
    int segments = seg_len > 0 ? (iters + seg_len - 1) / seg_len : 1;
    if (segments < 1) segments = 1;

    long long* d_prof = nullptr;
#ifdef SLICE_PROF
    const char* prof_path = getenv("PNP_SLICE_PROF");          // -DSLICE_PROF builds: per-phase clocks of each workgroup (single launch only)
#else
    const char* prof_path = nullptr;
#endif
    const size_t prof_n = (size_t)B * (2 + 6 * (size_t)iters);
    if (prof_path) {
        queues = 1; segments = 1;
        if (hipMalloc((void**)&d_prof, prof_n * sizeof(long long)) == hipSuccess) {
            (void)hipMemsetAsync(d_prof, 0, prof_n * sizeof(long long), s);
            a.prof = d_prof;
        }
    }
    // A batch larger than the chip runs in rounds (one slice per compute unit at a time), and the slices of the LAST round
    // are the ones the Infinity Cache still holds when the call returns: every other call walks the batch backwards, so that
    // a following call starts on warm data (DESIGN.md 4.1, round hand-over).  Slices are independent: results do not change.
    if (!prof_path && B > slice256_cus(f) && sch.slice_flip) { a.flip = f->flip; f->flip ^= 1; }
    hipError_t e = hipSuccess;
    if (queues > 1) {
        if (!f->ev_fork) e = hipEventCreateWithFlags(&f->ev_fork, hipEventDisableTiming);
        for (int q = 0; q < queues - 1 && e == hipSuccess; ++q) {
            if (f->side[q]) continue;
            e = hipStreamCreateWithFlags(&f->side[q], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_join[q], hipEventDisableTiming);
        }
        if (e == hipSuccess) e = hipEventRecord(f->ev_fork, s);
    }
    int c0 = 0;
    for (int q = 0; q < queues && e == hipSuccess; ++q) {
        const int Bq = (q == queues - 1) ? (B - c0) : (B / queues);
        hipStream_t sq = (q == 0) ? s : f->side[q - 1];
        if (q > 0) e = hipStreamWaitEvent(sq, f->ev_fork, 0);
        int done = 0;
        for (int g = 0; g < segments && e == hipSuccess; ++g) {
            const int n = (iters - done + (segments - g) - 1) / (segments - g);     // even split of what is left
            a.first = c0; a.B = Bq; a.iters = n;
            e = launch_slice(sq, a, prox);
            done += n;
        }
        if (q > 0 && e == hipSuccess) e = hipEventRecord(f->ev_join[q - 1], sq);
        if (q > 0 && e == hipSuccess) e = hipStreamWaitEvent(s, f->ev_join[q - 1], 0);
        c0 += Bq;
    }
    if (d_prof) {
        std::vector<long long> h(prof_n);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), d_prof, prof_n * sizeof(long long), hipMemcpyDeviceToHost);
        (void)hipFree(d_prof);
        if (FILE* fo = fopen(prof_path, "wb")) {
            const int hdr[2] = {B, iters};
            fwrite(hdr, sizeof(int), 2, fo);
            fwrite(h.data(), sizeof(long long), prof_n, fo);
            fclose(fo);
        }
    }
    return e;
}

}  // namespace pnp
