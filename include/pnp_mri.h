/*
 * pnp_mri.h -- C ABI of libpnpmri.so: the MI355X (gfx950) hot path of PnP-ADMM-CNC MRI
 * reconstruction.  Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * What it replaces.  zj15001/PNP_ADMM_CNC_MRI has no FFI on this path: the per-iteration loop is
 * ~15 lines of NumPy pasted inline into every solver function.  Each entry point below cites the
 * reference lines (aliases of SURVEY.md: S1 = "【1】ADMM_L1.py", S3 = "【3】PNP_ADMM_L1_D  .py",
 * S4 = "【4】ADMM_CNC .py", S6 = "【6】PNP_ADMM_CNC_D .py") it stands in for.  The only C-ABI
 * precedent in the reference is BM3D's ctypes layer (bm3d307/bm3d/bm3d_py.h:4-16,
 * bm3d_ctypes.py:194-240); INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (PNP_E_*); pnp_last_error() gives the
 *     thread-local message.  No exceptions, no aborts.
 *   - complex data = interleaved (re,im) float pairs ("complex64"), layout [B][H][W] row-major,
 *     k-space in un-shifted FFT order (DC at [0,0]), forward transform unnormalised, inverse
 *     carrying 1/(H*W) -- exactly np.fft.fft2 / np.fft.ifft2.
 *   - "dev" pointers are device pointers borrowed from the caller (never freed by the library);
 *     functions taking `on_device` accept either host or device memory.
 *   - all kernels are asynchronous on the ctx stream (pnp_set_stream); only pnp_sync,
 *     host-side downloads and pnp_timer_stop block.
 *   - one ctx per device per host thread; a ctx is not re-entrant.
 *   - hyper-parameters (alpha, lambda1, reo, b, thresholds) are C doubles, as the Python floats of
 *     the reference are; derived coefficients are formed in double and rounded to float once.
 *   - H, W in {256, 512}.  B <= Bmax.
 */
#ifndef PNP_MRI_H
#define PNP_MRI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNP_OK            0
#define PNP_E_ARG        -1   /* bad argument (null, size, unsupported H/W) */
#define PNP_E_HIP        -2   /* a HIP runtime call failed                  */
#define PNP_E_STATE      -3   /* call order (e.g. run before upload)        */
#define PNP_E_NOMEM      -4

#define PNP_ABI_VERSION   11

typedef struct pnp_ctx pnp_ctx;

/* ---- library / context ------------------------------------------------------------------ */
int         pnp_abi_version(void);
const char* pnp_last_error(void);
/* number of HIP devices visible (0 and PNP_OK when there is none). */
int         pnp_device_count(int* n);
/* What a measurement needs to be attributable to a card (bench.py's `config.device`): the shader clock the driver reports (MHz), the number of
 * compute units, the PCI bus id ("0000:05:00.0") and the architecture name ("gfx950...").  Strings are truncated to their buffers.  ABI 11. */
int         pnp_device_info(int device, int* clock_mhz, int* compute_units, char* pci_bus_id, int pci_len, char* arch, int arch_len);
/* CALIBRATION, not product: a streaming kernel with the slice-resident loop's ACCESS SHAPE and none of its arithmetic -- one 512-thread
 * workgroup per "slice" (256 KiB), 16 bytes per lane, eight accesses in flight per wave; per pass it reads z, w and a table and writes z, w
 * back (5 x 256 KiB) -- run back to back on `slices` slices for at least `seconds`; *gbs = bytes moved / HIP-event time.  What this card's
 * memory system gives such a kernel today: bench.py prints its own fraction of 8 TB/s next to the fraction of THIS number (boxes of a pool
 * differ by several per cent).  Allocates and frees 3 x slices x 256 KiB.  ABI 11. */
int         pnp_calibrate_stream(int device, int slices, double seconds, double* gbs);

int pnp_ctx_create(int device, int H, int W, int Bmax, pnp_ctx** out);
int pnp_ctx_destroy(pnp_ctx* ctx);
/* hipStream_t as void*; NULL = the default stream.  PnP solvers pass torch's current stream. */
int pnp_set_stream(pnp_ctx* ctx, void* hip_stream);
int pnp_sync(pnp_ctx* ctx);
/* 0 = generic kernels only, 1 = allow the fused gfx950 kernels where shapes permit (default). */
int pnp_set_fast_path(pnp_ctx* ctx, int enable);
/* Scheduling of the fused loops (results are bit-identical for every setting):
 *   queues          1..4 HIP queues the batch is split over (default 2; forked/joined on the ctx stream)
 *   mixed_launches  256x256 two-launch path: row workgroups of one half of a part share each launch with the column
 *                   workgroups of its other half (default 0: retired in round 2 -- measured within noise of plain row and
 *                   column launches on two queues; kept as a knob)
 *   chunk           >0: a queue runs all iterations on `chunk` slices before its next chunk; 0 (default): the path's own
 *                   default -- at 512x512 and for the double-precision 256x256 engine the chunks go round-robin to 4 queues
 *                   (16 resp. 24 slices each with queues >= 2; 48 resp. 96 on one queue), so the chunks in flight fit the
 *                   256 MiB Infinity Cache for a whole run; whole batch otherwise; <0: whole batch
 * Environment defaults read (once, range-checked: a malformed value fails pnp_ctx_create with PNP_E_ARG) at pnp_ctx_create:
 * PNP_FUSED_STREAMS, PNP_FUSED_SCHED, PNP_FUSED_CHUNK; PNP_SLICE (0 / 1), PNP_SLICE_MIN_B, PNP_SLICE_PAD_KB, PNP_SLICE_YH_PAD_KB,
 * PNP_FUSED_COLS.  The experiment knobs of the profiling scripts exist only in -DPNP_EXPERIMENT_KNOBS builds. */
int pnp_set_schedule(pnp_ctx* ctx, int queues, int mixed_launches, int chunk);
/* the schedule in force (any pointer may be NULL) */
int pnp_get_schedule(pnp_ctx* ctx, int* queues, int* mixed_launches, int* chunk);
/* what the next pnp_admm_*_run will do for the uploaded batch on the path in use: HIP queues, slices per chunk (the batch
 * size when it is not chunked), kernel launches per batched iteration (0: slice-resident, the iterations are a loop inside one
 * launch).  Any pointer may be NULL.  New in ABI 7; no counterpart in the reference. */
int pnp_get_plan(pnp_ctx* ctx, int* queues, int* chunk, int* launches_per_iteration);

/* ---- problem upload ---------------------------------------------------------------------- */
/* y: [B][H][W] complex64, the measurements  y = fft2(img)*mask + noises  (S4:102).
 * mask_bank: [K][H][W] uint8 in {0,1}, FFT-native layout (CS_MRI/Q_*.mat variable Q1; S4:185).
 * mask_id:   [B] int32 in [0,K): which mask each slice uses (NULL = all slices use mask 0);
 *            validated for host and device inputs alike (PNP_E_ARG when out of range).
 * Replaces the per-image `y`, `index = np.nonzero(mask)` set-up of S4:101-106.
 * A new problem (this call, pnp_synthesize_problem and their _f64 forms) INVALIDATES the ADMM state and x of the problem before it:
 * z, w and x are undefined until pnp_init_state, or pnp_set_state with BOTH z and w, has run -- the reference likewise builds z and w
 * anew per image (S4:103-109).  Until then every entry point that would read them (pnp_admm_l1_run, pnp_admm_cnc_run, pnp_get_state,
 * pnp_set_state with only one of z / w, and their _f64 forms) returns PNP_E_STATE. */
int pnp_upload_problem(pnp_ctx* ctx, const float* y, const uint8_t* mask_bank,
                       const int32_t* mask_id, int B, int K, int on_device);

/* On-device measurement synthesis (S4:102, "noise on all points"):
 * y = fft2(img)*mask + noise.  img: [B][H][W] float32; noise: [B][H][W] complex64 or, with
 * noise_per_slice = 0, one [H][W] complex64 array shared by all slices (the reference's
 * noises.mat).  Masks as in pnp_upload_problem.  Leaves y in the ctx like pnp_upload_problem. */
int pnp_synthesize_problem(pnp_ctx* ctx, const float* img, const float* noise, int noise_per_slice,
                           const uint8_t* mask_bank, const int32_t* mask_id, int B, int K,
                           int on_device);
/* copy the ctx's y back ([B][H][W] complex64). */
int pnp_download_y(pnp_ctx* ctx, float* y, int on_device);

/* z0 = |ifft2(y)| (complex modulus), w0 = 0                                     (S4:103-109) */
int pnp_init_state(pnp_ctx* ctx);
/* overwrite / read the ctx-owned ADMM state z, w ([B][H][W] float32 each); NULL skips one. */
int pnp_set_state(pnp_ctx* ctx, const float* z, const float* w, int on_device);
int pnp_get_state(pnp_ctx* ctx, float* z, float* w, int on_device);

/* Build now the per-problem tables the whole loops below will use (256x256 float contexts build them on the first loop call
 * otherwise: the slice-resident tables only when a loop really takes that path, so that step-wise PnP users never pay for
 * them).  Optional; benchmarks call it to keep table building out of a timed region without warm-up.  New in ABI 8. */
int pnp_prepare_loops(pnp_ctx* ctx);

/* ---- whole loops on the ctx-owned state (no host sync inside) ----------------------------- */
/* iters = 0: the loop body never runs; x is set to the current z (the reference's x = |ifft2(y)| = z0
 * straight after pnp_init_state, S4:103-107, 138).
 * ADMM_L1 main loop, S1:111-126:  x = dc(z,w); z = soft(x+w, reo*lambda1); w += x - z. */
int pnp_admm_l1_run(pnp_ctx* ctx, int iters, double lambda1, double reo);
/* ADMM_CNC main loop, S4:115-132: x = dc(z,w); s = soft(z,1/b);
 * t = (1-alpha) z + alpha (x+w) + alpha*reo*lambda1*b (z-s); z = soft(t, alpha*reo*lambda1);
 * w += x - z. */
int pnp_admm_cnc_run(pnp_ctx* ctx, int iters, double alpha, double lambda1, double reo, double b);
/* x of the last iteration ([B][H][W] float32) -- what the solvers return (S4:138). */
int pnp_download_x(pnp_ctx* ctx, float* x, int on_device);

/* ---- step-wise operators on caller-owned device pointers (the PnP path) ------------------- */
/* x-update / data-consistency solve, S4:119-124 == S6:266-271:
 *   X = fft2(z - w); X[mask] = (La2*X[mask] + y[mask])/(1+La2), La2 = 1/(2 reo);
 *   x = |Re ifft2(X)|.      z, w, x: [B][H][W] float32 device pointers (x may alias neither). */
int pnp_dc_step(pnp_ctx* ctx, const float* z_dev, const float* w_dev, float* x_dev, double reo);
/* z = soft(x + w, thr); w = w + x - z                                           (S1:123,126) */
int pnp_prox_l1_dual(pnp_ctx* ctx, const float* x_dev, float* z_dev, float* w_dev, double thr);
/* CNC z-update + dual update                                                 (S4:127-129,132) */
int pnp_prox_cnc_dual(pnp_ctx* ctx, const float* x_dev, float* z_dev, float* w_dev,
                      double alpha, double lambda1, double reo, double b);
/* t = (1-alpha) z + alpha (x+w) + alpha*reo*lambda1*b (z - s)                       (S6:301) */
int pnp_cnc_combine(pnp_ctx* ctx, const float* z_dev, const float* x_dev, const float* w_dev,
                    const float* s_dev, float* t_dev, double alpha, double lambda1, double reo, double b);
/* t = x + w  (the denoiser input of PNP_ADMM_L1_D, S3:290) */
int pnp_add(pnp_ctx* ctx, const float* a_dev, const float* b_dev, float* out_dev);
/* w = w + x - z, then x,z,w <- clamp(.,0,1) with torch.clamp_'s semantics: NaN stays NaN, +-inf go to the bounds, so a
 * non-finite denoiser output reaches the returned x instead of turning into 0                (S6:305-308) */
int pnp_dual_clamp(pnp_ctx* ctx, float* x_dev, float* z_dev, float* w_dev);

/* ---- operator API (batched; B slices of the ctx's H x W; complex64 device pointers) ------- */
int pnp_fft2_fwd(pnp_ctx* ctx, const float* in_dev, float* out_dev, int B);   /* np.fft.fft2  */
int pnp_fft2_inv(pnp_ctx* ctx, const float* in_dev, float* out_dev, int B);   /* np.fft.ifft2 */
/* A x = fft2(x) * mask (x real [B][H][W]); uses the uploaded masks.                 (S4:102) */
int pnp_A(pnp_ctx* ctx, const float* x_dev, float* k_dev);
/* A^H k = ifft2(k * mask) -> complex [B][H][W]                         (utils/utils.py:54) */
int pnp_AH(pnp_ctx* ctx, const float* k_dev, float* out_dev);
/* Df(x, mask, y) = ifft2(mask*fft2(x) - mask*y), x real -> complex    (utils/utils.py:50-55) */
int pnp_Df(pnp_ctx* ctx, const float* x_dev, float* out_dev);

/* ---- metrics (utils/utils_image.py:543-556, 622-636), per slice -------------------------- */
/* img_E = x*255 vs ground truth gt (uint8 [B][H][W]); writes psnr[b] (dB) and re[b].
 * x_dev = NULL means the ctx-owned x of the last pnp_admm_*_run. */
int pnp_metrics(pnp_ctx* ctx, const float* x_dev, const uint8_t* gt, int gt_on_device,
                double* psnr_host, double* re_host);

/* SSIM of img_E = x*255 vs gt per slice (utils/utils_image.py:570-615: Gaussian 11/1.5 window,
 * valid region, gray images), computed in double on the device.  x_dev = NULL: the ctx-owned x. */
int pnp_ssim(pnp_ctx* ctx, const float* x_dev, const uint8_t* gt, int gt_on_device, double* ssim_host);

/* ---- double-precision context -------------------------------------------------------------
 * The same loops (pnp_init_state, pnp_admm_l1_run, pnp_admm_cnc_run) with every buffer and every
 * arithmetic step in double / complex128: the reference itself runs in float64 (S4:109
 * `w = np.zeros(..., dtype=np.float64)`), and its committed CNC presets amplify fp32 round-off
 * ~1.08x per iteration, so this is the mode in which 100-iteration CNC runs meet 1e-5 end to end.
 * 256x256 runs on the fused two-launch kernels in double (a throughput path: ~2500 batched
 * iterations/s at 512 slices); other shapes on the generic kernels in double.
 * Since ABI 8 everything the whole-solver entry points need exists in double (synthesis, metrics, SSIM), so
 * ADMM_L1 / ADMM_CNC (precision='f64') run the reference's own arithmetic end to end; the step-wise / operator entry points
 * (PnP path: the reference itself switches to float32 there, S6:273-285) are float-only and return PNP_E_STATE on such a context. */
int pnp_ctx_create_f64(int device, int H, int W, int Bmax, pnp_ctx** out);
/* y: [B][H][W] complex128 (interleaved doubles); masks as pnp_upload_problem. */
int pnp_upload_problem_f64(pnp_ctx* ctx, const double* y, const uint8_t* mask_bank,
                           const int32_t* mask_id, int B, int K, int on_device);
/* y = fft2(img)*mask + noise in double (S4:102): img [B][H][W] float32 (the reference's img_L is float32,
 * utils/utils_image.py:181-182), noise complex128 [H][W] (noise_per_slice = 0: the reference's noises.mat) or [B][H][W].
 * The float32 image is widened exactly and transformed in double; NumPy >= 2 runs this one transform in complex64 before
 * promoting it, so the two y differ by that transform's float32 round-off (~1e-7 relative), which the committed CNC presets
 * amplify to ~2e-6 after 50 iterations -- inside the 1e-5 bar (tests/test_gpu_f64.py).  New in ABI 8. */
int pnp_synthesize_problem_f64(pnp_ctx* ctx, const float* img, const double* noise, int noise_per_slice,
                               const uint8_t* mask_bank, const int32_t* mask_id, int B, int K, int on_device);
/* copy the ctx's y back ([B][H][W] complex128).  New in ABI 8. */
int pnp_download_y_f64(pnp_ctx* ctx, double* y, int on_device);
int pnp_set_state_f64(pnp_ctx* ctx, const double* z, const double* w, int on_device);
int pnp_get_state_f64(pnp_ctx* ctx, double* z, double* w, int on_device);
int pnp_download_x_f64(pnp_ctx* ctx, double* x, int on_device);
int pnp_is_f64(pnp_ctx* ctx);
/* pnp_metrics / pnp_ssim on a double-precision context: img_E = x*255 formed in double from the float64 x, exactly as the
 * reference does (S4:139, 149-151).  x_dev = NULL: the ctx-owned x.  New in ABI 8. */
int pnp_metrics_f64(pnp_ctx* ctx, const double* x_dev, const uint8_t* gt, int gt_on_device,
                    double* psnr_host, double* re_host);
int pnp_ssim_f64(pnp_ctx* ctx, const double* x_dev, const uint8_t* gt, int gt_on_device, double* ssim_host);

/* ---- optional HIP backend of the denoisers' plain conv stacks ------------------------------
 * The north star keeps the CNN forward pass in PyTorch-ROCm; these two entry points are the opt-in `Denoiser(backend='hip')`
 * for the 64 -> 64 channel conv3x3 (+ bias, + ReLU) body layers of FFDNet / DnCNN / FDnCNN (models/network_ffdnet.py:58-73,
 * models/network_dncnn.py:36-67, models/basicblock.py:63-100 mode 'CR'), where configs 3-5 spend 99.8 % of their time:
 * an implicit GEMM on the fp32 matrix cores (exact f32 fma chains).  No ctx: caller-owned device tensors, any HIP stream.
 *   y = relu?( conv3x3(x, w) + bias + skip ),  stride 1, dilation d = 1..4, zero padding d (torch.nn.Conv2d(64, 64, 3, 1, d, dilation=d):
 *   d = 1 the plain stacks and DRUNet, d = 2..4 IRCNN's dilated layers, models/network_dncnn.py:87-101)
 *   x, y, skip: [n][H][W][64] float32 (NHWC); w_packed: 36 864 floats written by pnp_conv3x3_c64_pack (the kernel streams
 *   its weights from L2 in matrix-core fragment order); bias [64] or NULL; skip NULL or a tensor of y's shape added before the
 *   ReLU; y must alias neither x nor skip.  New in ABI 8. */
int pnp_conv3x3_c64_nhwc(void* hip_stream, const float* x_dev, const float* w_packed_dev, const float* bias_dev,
                         const float* skip_dev, float* y_dev, int n, int H, int W, int relu, int dilation);
/* w_oihw_dev: a torch.nn.Conv2d(64, 64, 3) weight, [64 out][64 in][3][3] contiguous -> w_packed_dev (36 864 floats).  Once
 * per model (again after the weights change). */
int pnp_conv3x3_c64_pack(void* hip_stream, const float* w_oihw_dev, float* w_packed_dev);
/* The same layer in SPLIT-HALF arithmetic on the f16 matrix cores ("f16x3"; csrc/kernels_conv_f16x3.hip): every float32 operand
 * is split into two halves, x = hi + lo / 2048, and a product becomes three v_mfma_f32_16x16x32_f16 (hi*hi, hi*lo, lo*hi: exact
 * products, float32 accumulation) -- float32-level results (operands carried to 2^-22, tests/test_gpu_conv.py measures the
 * distance from float64 beside the float32 kernel's) at several times the float32 matrix rate.  Same arguments and semantics as
 * pnp_conv3x3_c64_nhwc; w_packed from pnp_conv3x3_c64_pack_f16x3 (36 864 floats of storage, a different order: the two packings
 * are not interchangeable).  Operands must lie within the half range (|x|, |w| <= 65504): beyond it the result is inf / NaN.
 * New in ABI 9. */
int pnp_conv3x3_c64_nhwc_f16x3(void* hip_stream, const float* x_dev, const float* w_packed_dev, const float* bias_dev,
                               const float* skip_dev, float* y_dev, int n, int H, int W, int relu, int dilation);
int pnp_conv3x3_c64_pack_f16x3(void* hip_stream, const float* w_oihw_dev, float* w_packed_dev);
/* The f16x3 layer for C -> C channels, C a multiple of 64 up to 1024 (DRUNet's residual blocks at 128 / 256 / 512 channels,
 * models/network_unet.py:36-58 with models/basicblock.py:213-225): dilation 1, zero padding 1; x, y, skip [n][H][W][C] float32
 * (NHWC); w_packed: 9 C C floats of storage written by pnp_conv3x3_pack_f16x3 from a torch Conv2d(C, C, 3) weight; bias [C] or
 * NULL.  A workgroup computes 8 x 16 pixels x 64 output channels, its K loop runs over the C / 64 chunks of input channels; one
 * image must stay below 2 GiB (H W C floats).  C = 64 is the layer above (same packing).  New in ABI 9. */
int pnp_conv3x3_nhwc_f16x3(void* hip_stream, const float* x_dev, const float* w_packed_dev, const float* bias_dev,
                           const float* skip_dev, float* y_dev, int n, int C, int H, int W, int relu);
int pnp_conv3x3_pack_f16x3(void* hip_stream, const float* w_oihw_dev, float* w_packed_dev, int C);
/* The f16x3 layer with its tensors in the SPLIT ACTIVATION FORMAT, for chains of such layers (a plain stack's body, DRUNet's residual
 * blocks): a tensor keeps its shape and bytes -- [n][H][W][C], 256 bytes per pixel and block of 64 channels -- but a block holds
 * [64 hi halves][64 lo halves] of its 64 values, value = hi + lo / 2048 (the float32 result rounded to 2^-22 relative: what the
 * consuming layer's operands carry anyway).  The PRODUCING layer splits each output once in its epilogue; a consuming layer copies
 * 16-byte runs of halves into its operand tile instead of splitting every value of every tile again for every block of output
 * channels.  fmt: a mask saying which of x, skip, y are split (0 = all float32 = pnp_conv3x3_nhwc_f16x3); dilation 1..4 at C = 64, 1
 * otherwise.  New in ABI 10. */
#define PNP_FMT_X_SPLIT    1
#define PNP_FMT_SKIP_SPLIT 2
#define PNP_FMT_Y_SPLIT    4
int pnp_conv3x3_nhwc_f16x3_fmt(void* hip_stream, const float* x_dev, const float* w_packed_dev, const float* bias_dev,
                               const float* skip_dev, float* y_dev, int n, int C, int H, int W, int relu, int dilation, int fmt);
/* Which kernel the three f16x3 conv3x3 entry points above launch at dilation 1 (the results are bit-equal; tests and A/B measurements pin
 * one): -1 = by size (default: the WIDE kernel -- 16 x 16 pixel tiles, 64 x 64 wave tiles, compute + helper waves, kernels_conv_f16x3_wide.hip --
 * once every compute unit has two items of its own, the narrow one -- 8 x 16 tiles, two workgroups per unit -- below that), 0 = narrow always,
 * 1 = wide always.  Process-wide; returns the previous setting.  The environment variable PNP_CONV_WIDE sets the initial value.  New in ABI 11. */
int pnp_conv3x3_f16x3_set_variant(int variant);
/* pnp_conv3x3_tail_nchw (below) in the f16x3 arithmetic: x [n][H][W][64] (NHWC), w a torch Conv2d(64, cout, 3) weight (split inside the
 * kernel), 1 <= cout <= 4 -> y [n][cout][H][W] (NCHW), + bias.  On the vector units this layer costs as much as a 64 -> 64 layer of the
 * f16x3 kernel; as a 16-column matrix product it is bound by reading its input.  Same operand range as above.  New in ABI 9. */
int pnp_conv3x3_tail_nchw_f16x3(void* hip_stream, const float* x_nhwc_dev, const float* w_oihw_dev, const float* bias_dev,
                                float* y_nchw_dev, int n, int cout, int H, int W);
/* The same with a second input ADDED to x while it is staged (the U-Net's last skip sum, models/network_unet.py:134-135: the sum never
 * goes to memory).  New in ABI 10. */
int pnp_conv3x3_tail_add_nchw_f16x3(void* hip_stream, const float* x_nhwc_dev, const float* x2_nhwc_dev, const float* w_oihw_dev,
                                    const float* bias_dev, float* y_nchw_dev, int n, int cout, int H, int W);
/* DRUNet's scale changes in the f16x3 arithmetic (csrc/kernels_pix2x2_f16x3.hip), bias-free as in models/network_unet.py:95-107 with
 * models/basicblock.py:415-421, 439-445 -- with them `Denoiser(backend='hip_f16x3')` runs DRUNet without a MIOpen call:
 *   pnp_conv2x2s2_nhwc_f16x3    torch.nn.Conv2d(C, 2C, 2, 2, 0):           x [n][H][W][C] -> y [n][H/2][W/2][2C]  (H, W even; C = 64 k)
 *   pnp_convT2x2s2_nhwc_f16x3   torch.nn.ConvTranspose2d(C, C/2, 2, 2, 0): x [n][H][W][C] -> y [n][2H][2W][C/2]   (C = 128 k)
 * x2_dev: NULL, or a tensor of x's shape that is added to x while it is staged (the skip sums `m_up(x + x_skip)`,
 * models/network_unet.py:131-133).  w_packed from pnp_conv2x2_pack_f16x3 (transposed = 0: a Conv2d weight [2C][C][2][2], 8 C C floats
 * of storage; 1: a ConvTranspose2d weight [C][C/2][2][2], 2 C C floats).  Same operand range as the layers above.  New in ABI 10. */
int pnp_conv2x2s2_nhwc_f16x3(void* hip_stream, const float* x_dev, const float* x2_dev, const float* w_packed_dev, float* y_dev,
                             int n, int C, int H, int W);
int pnp_convT2x2s2_nhwc_f16x3(void* hip_stream, const float* x_dev, const float* x2_dev, const float* w_packed_dev, float* y_dev,
                              int n, int C, int H, int W);
int pnp_conv2x2_pack_f16x3(void* hip_stream, const float* w_dev, float* w_packed_dev, int C, int transposed);
/* First and last layer of the plain stacks (models/network_dncnn.py:52-62, models/network_ffdnet.py:50-56), direct convolutions:
 *   head: x [n][cin][H][W] (NCHW, 1 <= cin <= 8), w a torch Conv2d(cin, 64, 3) weight [64][cin][3][3] -> y [n][H][W][64] (NHWC), + bias, ReLU
 *   tail: x [n][H][W][64] (NHWC), w a torch Conv2d(64, cout, 3) weight [cout][64][3][3], 1 <= cout <= 4 -> y [n][cout][H][W] (NCHW), + bias
 * With them `Denoiser(backend='hip')` runs DnCNN / FDnCNN / FFDNet without a MIOpen call.  New in ABI 8. */
int pnp_conv3x3_head_nhwc(void* hip_stream, const float* x_nchw_dev, const float* w_oihw_dev, const float* bias_dev,
                          float* y_nhwc_dev, int n, int cin, int H, int W, int relu);
int pnp_conv3x3_tail_nchw(void* hip_stream, const float* x_nhwc_dev, const float* w_oihw_dev, const float* bias_dev,
                          float* y_nchw_dev, int n, int cout, int H, int W);
/* FFDNet's input and output stages folded into its first and last layer (models/network_ffdnet.py:58-73): no pad / pixel-unshuffle / concatenation
 * / pixel-shuffle / crop launches and no intermediate tensors around the stack.
 *   pnp_ffdnet_head_nhwc   x [n][1][h][w] (full resolution, any h, w >= 1: odd sizes are replicate-padded as the reference does), sigma_dev [n] or [1]
 *                          (sigma_per_image = 1 / 0): the layer's five input channels are the four pixel-unshuffled quarters of x and the noise
 *                          level; w a torch Conv2d(5, 64, 3) weight -> y [n][ceil(h/2)][ceil(w/2)][64] (NHWC), + bias, ReLU
 *   pnp_ffdnet_tail_f16x3  x [n][ceil(h/2)][ceil(w/2)][64] (NHWC), w a torch Conv2d(64, 4, 3) weight -> y [n][1][h][w]: the four output channels
 *                          written pixel-shuffled and cropped, in the f16x3 arithmetic of pnp_conv3x3_tail_nchw_f16x3.   New in ABI 10. */
int pnp_ffdnet_head_nhwc(void* hip_stream, const float* x_dev, const float* sigma_dev, int sigma_per_image, const float* w_oihw_dev,
                         const float* bias_dev, float* y_nhwc_dev, int n, int h, int w, int relu);
int pnp_ffdnet_tail_f16x3(void* hip_stream, const float* x_nhwc_dev, const float* w_oihw_dev, const float* bias_dev, float* y_dev, int n, int h, int w);
/* [n][64][H][W] <-> [n][H][W][64] (to_nhwc = 1 / 0): hand-over between PyTorch layers (NCHW) and the kernels above. */
int pnp_relayout_c64(void* hip_stream, const float* in_dev, float* out_dev, int n, int H, int W, int to_nhwc);

/* ---- timing on the ctx stream (HIP events) ------------------------------------------------ */
int pnp_timer_start(pnp_ctx* ctx);
int pnp_timer_stop(pnp_ctx* ctx, float* elapsed_ms);    /* records, synchronises, returns ms */

/* ---- introspection ------------------------------------------------------------------------ */
/* launches per batched ADMM iteration on the current path and schedule (0 = the iterations of a call are a
 * loop inside ONE launch), and which kernel family the loops take for the uploaded problem:
 *   "slice"   256x256 float, batches of >= 64 slices: one workgroup keeps a slice in registers for the whole run
 *   "fused"   two launches per iteration (256x256 float / double, 512x512 float)
 *   "generic" three launches per iteration (any H, W in {256, 512}; pnp_set_fast_path(ctx, 0)) */
int         pnp_kernels_per_iteration(pnp_ctx* ctx);
const char* pnp_path_name(pnp_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* PNP_MRI_H */
