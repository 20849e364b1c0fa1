// Internal declarations shared by the translation units of libpnpmri.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace pnp {

// complex / real types per precision: float2 for the production path, double2 for the fp64
// validation context (pnp_ctx_create_f64)
template <typename R> struct CxOf;
template <> struct CxOf<float>  { using type = float2; };
template <> struct CxOf<double> { using type = double2; };

// Scalars of the z/w update, pre-combined on the host in double and rounded once to R.
template <typename R>
struct ProxParamsT {
    R thr;      // L1: reo*lambda1            CNC: alpha*reo*lambda1   (outer soft threshold)
    R c1;       // CNC: 1-alpha
    R c2;       // CNC: alpha
    R c3;       // CNC: alpha*reo*lambda1*b
    R ib;       // CNC: 1/b   (inner clip level: z - soft(z,1/b) == clip(z,-1/b,1/b))
};
using ProxParams = ProxParamsT<float>;

enum RowIn  { IN_COMPLEX = 0, IN_REAL = 1, IN_REAL_DIFF = 2 };
enum RowEpi { EPI_COMPLEX = 0, EPI_ABS_REAL = 1, EPI_ABS_COMPLEX = 2, EPI_L1 = 3, EPI_CNC = 4 };
enum ColMid { MID_NONE = 0, MID_BLEND = 1, MID_MASK = 2, MID_RESID = 3, MID_MASK_ADD = 4 };

template <typename R>
struct RowArgsT {
    using C = typename CxOf<R>::type;
    const C* cin;      // IN_COMPLEX
    const R* rin0;     // IN_REAL / IN_REAL_DIFF (minuend)
    const R* rin1;     // IN_REAL_DIFF (subtrahend)
    C*       cout;     // EPI_COMPLEX
    R*       x_out;    // EPI_ABS_* (required) / EPI_L1, EPI_CNC (optional, may be null)
    R*       z;        // EPI_L1 / EPI_CNC: read old, write new
    R*       w;
    R        scale;    // applied to the transform output
    ProxParamsT<R> prox;
    int      nrows;    // B*H
};
using RowArgs = RowArgsT<float>;

template <typename R>
struct ColArgsT {
    using C = typename CxOf<R>::type;
    const C*       in;
    C*             out;        // may alias in
    const C*       y;          // MID_BLEND / MID_RESID: measurements; MID_MASK_ADD: noise
    const uint8_t* mask_bank;  // [K][H][W]
    const int32_t* mask_id;    // [B] or null
    R              c;          // MID_BLEND: 1/(1+La2)
    int            y_per_slice;// MID_MASK_ADD: 0 = one [H][W] noise array for all slices
    int            B;
};
using ColArgs = ColArgsT<float>;

// generic path (kernels_generic.hip); H, W in {256, 512}; R = float | double
template <typename R> hipError_t launch_rows(hipStream_t s, int W, RowIn in, bool inv, RowEpi epi, const RowArgsT<R>& a);
template <typename R> hipError_t launch_cols(hipStream_t s, int H, int W, bool pre_fwd, ColMid mid, bool post_inv, const ColArgsT<R>& a);
hipError_t upload_twiddles();       // fills the __device__ tables of the current device

// pointwise
hipError_t launch_prox(hipStream_t s, bool cnc, const float* x, float* z, float* w, ProxParams p, size_t n);
hipError_t launch_combine(hipStream_t s, const float* z, const float* x, const float* w, const float* sden,
                          float* t, float c1, float c2, float c3, size_t n);
hipError_t launch_add(hipStream_t s, const float* a, const float* b, float* o, size_t n);
hipError_t launch_dual_clamp(hipStream_t s, float* x, float* z, float* w, size_t n);
template <typename X> hipError_t launch_metrics(hipStream_t s, const X* x, const uint8_t* gt, double* acc /*[B][2]*/, int B, int N);
hipError_t upload_gauss();
template <typename X> hipError_t launch_ssim(hipStream_t s, const X* x, const uint8_t* gt, double* partial /*[B][tiles]*/, int B, int H, int W);
hipError_t launch_widen(hipStream_t s, const float* in, double* out, size_t n);      // float -> double, n % 4 == 0

// calibration (pnp_calibrate_stream): the slice-resident loop's access shape without its arithmetic, `passes` passes over `slices` slices of 256 KiB
hipError_t launch_calibrate_stream(hipStream_t s, float* z, float* w, const float* y, int slices, int passes);
// optional HIP backend of the denoisers' 64-channel conv3x3 body layers (kernels_conv.hip); activations NHWC float32
hipError_t launch_conv_pack_w(hipStream_t s, const float* w_oihw /*[64][64][3][3]*/, float* wfrag /*36 864 floats*/);
hipError_t launch_conv3x3_c64(hipStream_t s, const float* x, const float* wfrag, const float* bias, const float* skip,
                              float* y, int n, int H, int W, int relu, int dilation /* 1..4 */);
// the same layer in split-half arithmetic on the f16 matrix cores (kernels_conv_f16x3.hip): own packing, same buffer size
hipError_t launch_conv_pack_w_f16x3(hipStream_t s, const float* w_oihw /*[C][C][3][3]*/, float* wfrag /*9 C C floats*/, int C);
hipError_t launch_conv3x3_f16x3(hipStream_t s, const float* x, const float* wfrag, const float* bias, const float* skip,
                                float* y, int n, int C /* 64 k <= 1024 */, int H, int W, int relu, int dilation /* 1..4; 1 if C > 64 */,
                                int fmt = 0 /* bit 0: x, bit 1: skip, bit 2: y in the split activation format (f16x3_common.h) */);
// the same layer at dilation 1 with 64 x 64 wave tiles, compute + helper waves (kernels_conv_f16x3_wide.hip): same arguments, same packed
// weights, bit-equal results; launch_conv3x3_f16x3 dispatches to it (conv_wide_mode below) -- callers never name it
hipError_t launch_conv3x3_f16x3_wide(hipStream_t s, const float* x, const float* wfrag, const float* bias, const float* skip,
                                     float* y, int n, int C, int H, int W, int relu, int fmt);
int conv_wide_mode();            // -1 = by size, 0 = never, 1 = always (dilation 1); initial value: PNP_CONV_WIDE; pnp_conv3x3_f16x3_set_variant
int conv_set_wide_mode(int m);   // returns the previous setting
hipError_t launch_conv3x3_tail_f16x3(hipStream_t s, const float* x_nhwc, const float* x2_nhwc /* null or added to x */, const float* w_oihw,
                                     const float* bias, float* y_nchw, int n, int cout, int H, int W,
                                     int shuffle_h = 0, int shuffle_w = 0 /* FFDNet: cout = 4 written as one pixel-shuffled [shuffle_h][shuffle_w] channel */);
hipError_t launch_ffdnet_head(hipStream_t s, const float* x_full /* [n][1][h][w] */, const float* sigma, int sigma_per_image, const float* w_oihw /* [64][5][3][3] */,
                              const float* bias, float* y_nhwc /* [n][ceil(h/2)][ceil(w/2)][64] */, int n, int h, int w, int relu);
// DRUNet's 2 x 2 stride-2 convolution (C -> 2C, up = 0) and 2 x 2 transposed convolution (C -> C/2, up = 1) in the same arithmetic
// (kernels_pix2x2_f16x3.hip); x2: null or a tensor of x's shape added to it
hipError_t launch_pix2_pack_w_f16x3(hipStream_t s, const float* w, float* wfrag /* 8 C C floats (down), 2 C C (up) */, int C, int up);
hipError_t launch_pix2x2_f16x3(hipStream_t s, const float* x, const float* x2, const float* wfrag, float* y, int n, int C, int H, int W, int up);
hipError_t launch_relayout64(hipStream_t s, const float* in, float* out, int n, int HW, bool to_nhwc);
hipError_t launch_conv3x3_head(hipStream_t s, const float* x_nchw, const float* w_oihw, const float* bias, float* y_nhwc,
                               int n, int cin, int H, int W, int relu);
hipError_t launch_conv3x3_tail(hipStream_t s, const float* x_nhwc, const float* w_oihw, const float* bias, float* y_nchw,
                               int n, int cout, int H, int W);

// How the fused loops are scheduled (scheduling only: results are bit-identical for every setting).
struct FusedSchedule {
    int queues = 2;         // HIP queues the batch is split over (1..4); kernel heads/tails overlap
    int mixed = 0;          // 256x256: row workgroups of one half + column workgroups of the other per launch (k_fmixed)
    int chunk = 0;          // >0: a queue finishes all iterations on `chunk` slices before its next chunk; 0: the path's default
                            // (chunk_plan below); <0: off (whole batch / two halves)
    int l1_two_state = 0;   // test hook: ADMM_L1 keeps z and w every iteration instead of u only
    // Experiment knobs.  They stay at these defaults unless the library is built with -DPNP_EXPERIMENT_KNOBS (profiles/variants.sh
    // does); such a build reads them from the environment ONCE, at pnp_ctx_create (api.hip, read_knobs), range-checked.
    int chunk_queues = 0;   // >0: queues of the chunked schedules           (PNP_F512_QUEUES / PNP_F256S_QUEUES)
    int slice_xor = 0;      // slice <-> workgroup permutation b ^ xor          (PNP_SLICE_XOR)
    int slice_queues = 1;   // slice-resident run cut over HIP queues ...       (PNP_SLICE_QUEUES)
    int slice_segment = 0;  // ... and into launches of this many iterations    (PNP_SLICE_SEGMENT)
    int slice_flip = 1;     // every other multi-round call walks the batch backwards (PNP_SLICE_FLIP)
};

// Chunked schedules of the 512x512 loops and of the split-chain (double) 256x256 loops: a queue runs ALL iterations of a run on
// `chunk` slices before its next chunk, and the chunks go round-robin to Q queues -- Q chunks in flight, whose working set
// (Q * chunk * 4 MiB at 512x512, * 2.5 MiB in double) stays around the 256 MiB Infinity Cache and whose kernel tails overlap
// each other's heads.  Measured on one box each (it/s; profiles/bench_r02):
//   512x512, 256 slices: whole batch 1057-1171 (a slow mode on some boxes), 1 x 48: 1161-1172, 2 x 32: 1338, 3 x 24: 1360,
//                        4 x 16: 1358, 4 x 24: 1361, 4 x 8: 1244
//   double, 512 slices:  two halves on two queues 2186-2400, 1 x 96: 2423, 2 x 48: 2563, 3 x 32: 2570, 4 x 24: 2573, 4 x 64: 2436
struct ChunkPlan {
    int queues, chunk;      // chunk == B and queues == 1: the whole batch at once
};
static inline ChunkPlan chunk_plan(int B, const FusedSchedule& sch, bool is512, bool is_double, int env_queues /* <=0: none */) {
    ChunkPlan p;
    p.queues = env_queues > 0 ? env_queues : (sch.queues >= 2 ? 4 : 1);
    if (p.queues > 4) p.queues = 4;
    const int dflt = is512 ? (p.queues >= 2 ? 16 : 48) : (is_double ? (p.queues >= 2 ? 24 : 96) : 0);
    p.chunk = sch.chunk != 0 ? sch.chunk : dflt;
    if (p.chunk <= 0) {                                     // off
        if (!is512 && sch.queues >= 2 && B >= 64) { p.queues = 2; p.chunk = ((B / 2) + 1) & ~1; }     // two halves (round 1)
        else { p.queues = 1; p.chunk = B; }
    }
    p.chunk &= ~1;
    if (p.chunk < 2) p.chunk = 2;
    if (p.chunk >= B) { p.chunk = B > 2 ? ((B + 1) & ~1) : 2; p.queues = 1; }
    return p;
}
static inline int chunk_plan_launches(int B, const ChunkPlan& p) { return 2 * ((B + p.chunk - 1) / p.chunk); }

// fused 256x256 path (kernels_fused256.hip): state resident in the ctx, two slices packed into
// one complex transform.  See DESIGN.md.
struct Fused256;
Fused256*  fused256_create(int Bmax, hipError_t* err);
void       fused256_destroy(Fused256*);
// builds the Hermitian-symmetrised measurement / mask tables from y and the masks
hipError_t fused256_prepare(Fused256*, hipStream_t s, const float2* y, const uint8_t* mask_bank,
                            const int32_t* mask_id, int B);
// iters iterations of x=dc(z,w); (z,w)=prox(x,z,w) on z,w [B][256][256]; x written on the last
hipError_t fused256_run(Fused256*, hipStream_t s, float* z, float* w, float* x, int B, int iters,
                        bool cnc, float dc_c, ProxParams p, const FusedSchedule& sch);
// one data-consistency step on caller pointers
hipError_t fused256_dc(Fused256*, hipStream_t s, const float* z, const float* w, float* x, int B, float dc_c);

// "split chain" engine for 256x256 (kernels_fused256.hip, k_fcols2): one column chain per thread and
// slice; R = float | double.  y is [B][256][256] complex in R (float2 / double2 layout).
template <typename R> struct Fused256S;
template <typename R> Fused256S<R>* fused256s_create(int Bmax, hipError_t* err);
template <typename R> void          fused256s_destroy(Fused256S<R>*);
template <typename R> hipError_t    fused256s_prepare(Fused256S<R>*, hipStream_t s, const void* y, const uint8_t* mask_bank,
                                                      const int32_t* mask_id, int B);
template <typename R> hipError_t    fused256s_run(Fused256S<R>*, hipStream_t s, R* z, R* w, R* x, int B, int iters, bool cnc,
                                                  R dc_c, ProxParamsT<R> p, const FusedSchedule& sch);
template <typename R> hipError_t    fused256s_dc(Fused256S<R>*, hipStream_t s, const R* z, const R* w, R* x, int B, R dc_c);

// slice-resident 256x256 path (kernels_slice256.hip): one workgroup keeps one slice in registers for a whole run
struct Slice256;
Slice256*  slice256_create(int Bmax, int pad_kb /* state */, int yh_pad_kb /* table */, hipError_t* err);
void       slice256_destroy(Slice256*);
int        slice256_cus(const Slice256*);     // compute units of the device (= slices in flight)
hipError_t slice256_prepare(Slice256*, hipStream_t s, const float2* y, const uint8_t* mask_bank, const int32_t* mask_id, int B);
// z, w in SLICE ORDER (slice_layout.h, sl_state_index); x comes out in natural order
hipError_t slice256_run(Slice256*, hipStream_t s, float* z, float* w, float* x, int B, int iters, bool cnc, float dc_c,
                        ProxParams p, const FusedSchedule& sch);
// in-place conversion of both state arrays [B][256][256] between natural order and slice order
hipError_t slice256_state_order(Slice256*, hipStream_t s, float* z, float* w, int B, bool to_slice);

// fused 512x512 path (kernels_fused512.hip): same scheme with 32-lane transforms
struct Fused512;
Fused512*  fused512_create(int Bmax, hipError_t* err);
void       fused512_destroy(Fused512*);
hipError_t fused512_prepare(Fused512*, hipStream_t s, const float2* y, const uint8_t* mask_bank,
                            const int32_t* mask_id, int B);
hipError_t fused512_run(Fused512*, hipStream_t s, float* z, float* w, float* x, int B, int iters,
                        bool cnc, float dc_c, ProxParams p, const FusedSchedule& sch);
hipError_t fused512_dc(Fused512*, hipStream_t s, const float* z, const float* w, float* x, int B, float dc_c);

}  // namespace pnp
