// C ABI of libpnpmri.so (include/pnp_mri.h): context, problem upload, whole ADMM loops, step-wise
// operators.  Host-side only; kernels live in kernels_generic.hip / kernels_fused256.hip.
#include "../../include/pnp_mri.h"
#include "internal.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <dlfcn.h>
#include <new>
#include <vector>

using namespace pnp;

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                               \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(PNP_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// roctx ranges around the loops so that rocprofv3 --marker-trace output is self-describing.  The
// marker library is looked up at run time (it is part of the ROCm image, not a link dependency);
// without it the ranges are no-ops.
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            void* h = dlopen(name, RTLD_LAZY | RTLD_LOCAL);
            if (!h) continue;
            push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
            pop = (int (*)())dlsym(h, "roctxRangePop");
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
struct Range {
    static Roctx& api() { static Roctx r; return r; }
    explicit Range(const char* name) { if (api().push) api().push(name); }
    ~Range() { if (api().pop) api().pop(); }
};
}  // namespace

struct pnp_ctx {
    int device = 0, H = 0, W = 0, Bmax = 0;
    int B = 0, K = 0;                 // current problem (0 = none uploaded)
    size_t N = 0;                     // H*W
    hipStream_t stream = nullptr;
    bool fast = true;
    bool have_x = false;
    bool have_state = false;         // z / w hold a defined state for the CURRENT problem: cleared by upload / synthesize, set by pnp_init_state or pnp_set_state(z, w)
    float2* y = nullptr;              // [Bmax][H][W]
    float2* work = nullptr;           // [Bmax][H][W] transform intermediate
    float *z = nullptr, *w = nullptr, *x = nullptr;
    uint8_t* mask_bank = nullptr;     // [Kcap][H][W]
    int Kcap = 0;
    int32_t* mask_id = nullptr;       // [Bmax]
    uint8_t* gt = nullptr;            // [Bmax][H][W] (metrics, lazily)
    double* acc = nullptr;            // [Bmax][2]
    double* ssim_part = nullptr;      // [Bmax][tiles] (lazily)
    void* stage = nullptr;            // staging for host inputs of synthesize
    size_t stage_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    Fused256* fused = nullptr;        // 256 x 256
    Fused512* fused5 = nullptr;       // 512 x 512
    Slice256* slice = nullptr;        // 256 x 256 slice-resident loops (whole runs; pnp_dc_step stays on `fused`)
    bool slice_ready = false;        // the loops of the uploaded problem take the slice-resident path ...
    bool slice_tabs = false;         // ... and its tables have been built (on the first such loop, or by pnp_prepare_loops)
    bool state_sliced = false;       // c->z / c->w are in the slice-resident kernel's order (slice_layout.h, sl_state_index)
    int slice_min_b = 0;              // batches at least this large run their loops slice-resident
    bool slice_force = false;
    Fused256S<float>* fs32 = nullptr;   // 256 x 256 "split chain" engine in float (PNP_FUSED_COLS=2) ...
    Fused256S<double>* fs64 = nullptr;  // ... and in double: the fast path of an fp64 context
    FusedSchedule sched;              // defaults overridable by PNP_FUSED_* (read at creation) / pnp_set_schedule
    bool fused_ready = false;         // tables prepared for the current problem
    bool fused_tabs = false;          // 256x256 float: the two-launch tables themselves (built on first use when the slice-resident tables serve the loops)
    // fp64 validation context (pnp_ctx_create_f64): same loop, generic kernels, double buffers
    bool f64 = false;
    double2* yd = nullptr;
    double2* workd = nullptr;
    double *zd = nullptr, *wd = nullptr, *xd = nullptr;
};

static bool supported(int n) { return n == 256 || n == 512; }

// Environment knobs: read ONCE per context (pnp_ctx_create), whole-string integers, range-checked -- a stray or mistyped
// variable fails the creation with PNP_E_ARG instead of silently changing which kernel a caller gets.
struct Knobs {
    int slice = -1;            // PNP_SLICE: 0 never, 1 always, unset: batches of at least slice_min_b slices
    int slice_min_b = 64;      // PNP_SLICE_MIN_B
    int slice_pad_kb = 4;      // PNP_SLICE_PAD_KB / PNP_SLICE_YH_PAD_KB: padding per slice of the slice path's own arrays
    int slice_yh_pad_kb = 4;
    int fused_cols = 1;        // PNP_FUSED_COLS: 2 = the split-chain column kernel in float
};
static int knob(const char* name, int lo, int hi, int* out) {
    const char* e = getenv(name);
    if (!e) return PNP_OK;
    char* end = nullptr;
    const long v = strtol(e, &end, 10);
    if (end == e || *end != '\0' || v < lo || v > hi)
        return fail(PNP_E_ARG, "pnp_ctx_create: environment variable %s=\"%s\" is not an integer in [%d, %d]", name, e, lo, hi);
    *out = (int)v;
    return PNP_OK;
}
static int read_knobs(Knobs* k, FusedSchedule* sch) {
    int rc;
    if ((rc = knob("PNP_SLICE", 0, 1, &k->slice))) return rc;
    if ((rc = knob("PNP_SLICE_MIN_B", 1, 1 << 20, &k->slice_min_b))) return rc;
    if ((rc = knob("PNP_SLICE_PAD_KB", 0, 64, &k->slice_pad_kb))) return rc;
    if ((rc = knob("PNP_SLICE_YH_PAD_KB", 0, 64, &k->slice_yh_pad_kb))) return rc;
    if ((rc = knob("PNP_FUSED_COLS", 1, 2, &k->fused_cols))) return rc;
    if ((rc = knob("PNP_FUSED_STREAMS", 1, 4, &sch->queues))) return rc;
    if ((rc = knob("PNP_FUSED_SCHED", 0, 1, &sch->mixed))) return rc;
    if ((rc = knob("PNP_FUSED_CHUNK", -1, 1 << 20, &sch->chunk))) return rc;
    if ((rc = knob("PNP_FUSED_L1_TWO_STATE", 0, 1, &sch->l1_two_state))) return rc;     // test hook
#ifdef PNP_EXPERIMENT_KNOBS
    // A/B builds only (profiles/variants.sh): the product library does not look at these variables
    if ((rc = knob("PNP_F512_QUEUES", 1, 4, &sch->chunk_queues))) return rc;
    if ((rc = knob("PNP_F256S_QUEUES", 1, 4, &sch->chunk_queues))) return rc;
    if ((rc = knob("PNP_SLICE_XOR", 0, 1 << 20, &sch->slice_xor))) return rc;
    if ((rc = knob("PNP_SLICE_QUEUES", 1, 4, &sch->slice_queues))) return rc;
    if ((rc = knob("PNP_SLICE_SEGMENT", 0, 1 << 20, &sch->slice_segment))) return rc;
    if ((rc = knob("PNP_SLICE_FLIP", 0, 1, &sch->slice_flip))) return rc;
#endif
    return PNP_OK;
}

static ProxParams make_prox_l1(double lambda1, double reo) {
    ProxParams p{};
    p.thr = (float)(reo * lambda1);
    return p;
}
static ProxParams make_prox_cnc(double alpha, double lambda1, double reo, double b) {
    ProxParams p{};
    p.thr = (float)(alpha * reo * lambda1);
    p.c1 = (float)(1.0 - alpha);
    p.c2 = (float)alpha;
    p.c3 = (float)(alpha * reo * lambda1 * b);
    p.ib = (float)(1.0 / b);
    return p;
}
static float dc_coeff(double reo) { return (float)(1.0 / (1.0 + 1.0 / 2.0 / reo)); }

// The slice-resident loops keep z / w in their own order; everything else (the other kernel families, pnp_get_state /
// pnp_set_state, pnp_init_state) sees natural [H][W].  One kernel converts when the need changes (into the slice path's own
// padded arrays; in place with PNP_SLICE_PAD_KB=0).
static int state_order(pnp_ctx* c, bool sliced) {
    if (c->state_sliced == sliced) return PNP_OK;
    if (c->B > 0) {
        hipError_t e = slice256_state_order(c->slice, c->stream, c->z, c->w, c->B, sliced);
        if (e != hipSuccess) return fail(PNP_E_HIP, "state order: %s", hipGetErrorString(e));
    }
    c->state_sliced = sliced;
    return PNP_OK;
}

static bool use_fused(pnp_ctx* c) { return c->fast && (c->fused || c->fused5 || c->fs32 || c->fs64) && c->fused_ready; }

extern "C" {

int pnp_abi_version(void) { return PNP_ABI_VERSION; }
const char* pnp_last_error(void) { return g_err; }

int pnp_device_count(int* n) {
    if (!n) return fail(PNP_E_ARG, "pnp_device_count: null");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { (void)hipGetLastError(); c = 0; }
    *n = c;
    return PNP_OK;
}

int pnp_device_info(int device, int* clock_mhz, int* compute_units, char* pci_bus_id, int pci_len, char* arch, int arch_len) {
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, device));
    if (clock_mhz) *clock_mhz = p.clockRate / 1000;
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (pci_bus_id && pci_len > 0) { pci_bus_id[0] = 0; HIPCHK(hipDeviceGetPCIBusId(pci_bus_id, pci_len, device)); }
    if (arch && arch_len > 0) { snprintf(arch, (size_t)arch_len, "%s", p.gcnArchName); }
    return PNP_OK;
}

int pnp_calibrate_stream(int device, int slices, double seconds, double* gbs) {
    if (!gbs || slices < 1 || slices > 4096 || !(seconds > 0.0) || seconds > 30.0) return fail(PNP_E_ARG, "pnp_calibrate_stream: 1 <= slices <= 4096, 0 < seconds <= 30, gbs non-null");
    HIPCHK(hipSetDevice(device));
    const size_t bytes = (size_t)slices << 18;
    float *z = nullptr, *w = nullptr, *y = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = PNP_OK;
    double moved = 0.0, ms_total = 0.0;
#define CAL(x) do { if (rc == PNP_OK) { hipError_t e_ = (x); if (e_ != hipSuccess) rc = fail(PNP_E_HIP, "pnp_calibrate_stream: %s: %s", #x, hipGetErrorString(e_)); } } while (0)
    CAL(hipMalloc((void**)&z, bytes)); CAL(hipMalloc((void**)&w, bytes)); CAL(hipMalloc((void**)&y, bytes));
    CAL(hipMemset(z, 0, bytes)); CAL(hipMemset(w, 0, bytes)); CAL(hipMemset(y, 0, bytes));
    CAL(hipEventCreate(&e0)); CAL(hipEventCreate(&e1));
    const int passes = 50;
    CAL(launch_calibrate_stream(nullptr, z, w, y, slices, 5));                           // warm-up
    CAL(hipDeviceSynchronize());
    while (rc == PNP_OK && ms_total < seconds * 1e3) {
        CAL(hipEventRecord(e0, nullptr));
        CAL(launch_calibrate_stream(nullptr, z, w, y, slices, passes));
        CAL(hipEventRecord(e1, nullptr));
        CAL(hipEventSynchronize(e1));
        float ms = 0.f;
        CAL(hipEventElapsedTime(&ms, e0, e1));
        ms_total += ms;
        moved += 5.0 * 262144.0 * slices * passes;
    }
#undef CAL
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (z) (void)hipFree(z);
    if (w) (void)hipFree(w);
    if (y) (void)hipFree(y);
    if (rc == PNP_OK) *gbs = moved / (ms_total * 1e-3) / 1e9;
    return rc;
}

static int ctx_create_any(int device, int H, int W, int Bmax, pnp_ctx** out, bool f64) {
    if (!out) return fail(PNP_E_ARG, "pnp_ctx_create: out is null");
    *out = nullptr;
    if (!supported(H) || !supported(W)) return fail(PNP_E_ARG, "pnp_ctx_create: H, W must be 256 or 512 (got %dx%d)", H, W);
    if (Bmax < 1) return fail(PNP_E_ARG, "pnp_ctx_create: Bmax must be >= 1");
    Knobs kn;
    FusedSchedule sched0;
    if (int rk = read_knobs(&kn, &sched0)) return rk;
    HIPCHK(hipSetDevice(device));
    pnp_ctx* c = new (std::nothrow) pnp_ctx();
    if (!c) return fail(PNP_E_NOMEM, "pnp_ctx_create: host allocation failed");
    c->device = device; c->H = H; c->W = W; c->Bmax = Bmax; c->N = (size_t)H * W; c->f64 = f64;
    c->sched = sched0;
    const size_t BN = (size_t)Bmax * c->N;
    hipError_t e = hipSuccess;
    auto alloc = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    if (f64) {
        alloc((void**)&c->yd, BN * sizeof(double2));
        alloc((void**)&c->workd, BN * sizeof(double2));
        alloc((void**)&c->zd, BN * sizeof(double));
        alloc((void**)&c->wd, BN * sizeof(double));
        alloc((void**)&c->xd, BN * sizeof(double));
    } else {
        alloc((void**)&c->y, BN * sizeof(float2));
        alloc((void**)&c->work, BN * sizeof(float2));
        alloc((void**)&c->z, BN * sizeof(float));
        alloc((void**)&c->w, BN * sizeof(float));
        alloc((void**)&c->x, BN * sizeof(float));
    }
    alloc((void**)&c->mask_id, (size_t)Bmax * sizeof(int32_t));
    alloc((void**)&c->acc, (size_t)Bmax * 2 * sizeof(double));
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) e = upload_twiddles();
    if (e == hipSuccess) e = upload_gauss();
    if (e != hipSuccess) {
        pnp_ctx_destroy(c);
        return fail(e == hipErrorOutOfMemory ? PNP_E_NOMEM : PNP_E_HIP, "pnp_ctx_create: %s", hipGetErrorString(e));
    }
    if (H == W && (!f64 || H == 256)) {
        hipError_t fe = hipSuccess;
        if (f64)                                         c->fs64 = fused256s_create<double>(Bmax, &fe);
        else if (H == 256 && kn.fused_cols == 2)         c->fs32 = fused256s_create<float>(Bmax, &fe);
        else if (H == 256) {
            c->fused = fused256_create(Bmax, &fe);
            // Slice-resident loops: one workgroup (= one compute unit) per slice, so they pay off once the batch
            // fills the chip; small batches stay on the two-launch path, which spreads a slice over many CUs.
            // PNP_SLICE=0 never, =1 always, unset: batches of at least PNP_SLICE_MIN_B slices (slice_pays()).
            const int mode = kn.slice;
            c->slice_min_b = mode == 1 ? 1 : kn.slice_min_b;
            c->slice_force = (mode == 1);
            if (c->fused && mode != 0 && Bmax >= c->slice_min_b) {
                hipError_t se = hipSuccess;
                c->slice = slice256_create(Bmax, kn.slice_pad_kb, kn.slice_yh_pad_kb, &se);
                // no room for the slice-resident tables (256 KiB per slice on top of the two-launch tables): the context
                // degrades to the two-launch path (pnp_path_name says "fused") -- unless the caller forced PNP_SLICE=1
                if (!c->slice) {
                    (void)hipGetLastError();
                    if (mode == 1) { fe = se; fused256_destroy(c->fused); c->fused = nullptr; }
                }
            }
        }
        else                                             c->fused5 = fused512_create(Bmax, &fe);
        if (!c->fused && !c->fused5 && !c->fs32 && !c->fs64) {
            pnp_ctx_destroy(c);
            return fail(PNP_E_HIP, "pnp_ctx_create: fused path: %s", hipGetErrorString(fe));
        }
    }
    *out = c;
    return PNP_OK;
}

int pnp_ctx_create(int device, int H, int W, int Bmax, pnp_ctx** out) { return ctx_create_any(device, H, W, Bmax, out, false); }
int pnp_ctx_create_f64(int device, int H, int W, int Bmax, pnp_ctx** out) { return ctx_create_any(device, H, W, Bmax, out, true); }

int pnp_ctx_destroy(pnp_ctx* c) {
    if (!c) return PNP_OK;
    (void)hipSetDevice(c->device);
    if (c->fused) fused256_destroy(c->fused);
    if (c->fused5) fused512_destroy(c->fused5);
    if (c->slice) slice256_destroy(c->slice);
    if (c->fs32) fused256s_destroy(c->fs32);
    if (c->fs64) fused256s_destroy(c->fs64);
    void* ptrs[] = {c->y, c->work, c->z, c->w, c->x, c->mask_bank, c->mask_id, c->gt, c->acc, c->stage, c->ssim_part,
                    c->yd, c->workd, c->zd, c->wd, c->xd};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    delete c;
    return PNP_OK;
}

#define CTX(c) do { if (!(c)) return fail(PNP_E_ARG, "%s: ctx is null", __func__); HIPCHK(hipSetDevice((c)->device)); } while (0)
#define NEED_PROBLEM(c) do { if ((c)->B <= 0) return fail(PNP_E_STATE, "%s: no problem uploaded", __func__); } while (0)
#define NEED_STATE(c) do { if (!(c)->have_state) return fail(PNP_E_STATE, "%s: z / w are undefined since the last upload / synthesize (call pnp_init_state or pnp_set_state with both z and w)", __func__); } while (0)
#define F32_ONLY(c) do { if ((c)->f64) return fail(PNP_E_STATE, "%s: not available on an fp64 validation context", __func__); } while (0)
#define F64_ONLY(c) do { if (!(c)->f64) return fail(PNP_E_STATE, "%s: needs a context made by pnp_ctx_create_f64", __func__); } while (0)

int pnp_set_stream(pnp_ctx* c, void* s) { CTX(c); c->stream = (hipStream_t)s; return PNP_OK; }
int pnp_sync(pnp_ctx* c) { CTX(c); HIPCHK(hipStreamSynchronize(c->stream)); return PNP_OK; }
int pnp_set_fast_path(pnp_ctx* c, int enable) { CTX(c); c->fast = enable != 0; return PNP_OK; }
int pnp_get_schedule(pnp_ctx* c, int* queues, int* mixed_launches, int* chunk) {
    CTX(c);
    if (queues) *queues = c->sched.queues;
    if (mixed_launches) *mixed_launches = c->sched.mixed;
    if (chunk) *chunk = c->sched.chunk;
    return PNP_OK;
}
int pnp_set_schedule(pnp_ctx* c, int queues, int mixed_launches, int chunk) {
    CTX(c);
    if (queues < 1 || queues > 4) return fail(PNP_E_ARG, "pnp_set_schedule: queues in 1..4");
    c->sched.queues = queues; c->sched.mixed = mixed_launches != 0; c->sched.chunk = chunk;      // the other fields keep their values
    return PNP_OK;
}

static int copy_in(pnp_ctx* c, void* dst, const void* src, size_t bytes, int on_device) {
    HIPCHK(hipMemcpyAsync(dst, src, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    if (!on_device) HIPCHK(hipStreamSynchronize(c->stream));   // caller may reuse its host buffer
    return PNP_OK;
}
static int copy_out(pnp_ctx* c, void* dst, const void* src, size_t bytes, int on_device) {
    HIPCHK(hipMemcpyAsync(dst, src, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    if (!on_device) HIPCHK(hipStreamSynchronize(c->stream));
    return PNP_OK;
}

static int set_masks(pnp_ctx* c, const uint8_t* mask_bank, const int32_t* mask_id, int B, int K, int on_device) {
    if (!mask_bank) return fail(PNP_E_ARG, "mask_bank is null");
    if (B < 1 || B > c->Bmax) return fail(PNP_E_ARG, "B=%d out of range [1,%d]", B, c->Bmax);
    if (K < 1) return fail(PNP_E_ARG, "K must be >= 1");
    if (K > c->Kcap) {
        if (c->mask_bank) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->mask_bank)); c->mask_bank = nullptr; c->Kcap = 0; }
        HIPCHK(hipMalloc((void**)&c->mask_bank, (size_t)K * c->N));
        c->Kcap = K;
    }
    int rc = copy_in(c, c->mask_bank, mask_bank, (size_t)K * c->N, on_device);
    if (rc) return rc;
    if (mask_id) {
        // ids index the mask bank inside the kernels: validate them for host AND device inputs
        // (B int32 values; a device array is read back once -- problem upload is not the hot path)
        std::vector<int32_t> tmp;
        const int32_t* ids = mask_id;
        if (on_device) {
            tmp.resize((size_t)B);
            HIPCHK(hipMemcpyAsync(tmp.data(), mask_id, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            ids = tmp.data();
        }
        for (int i = 0; i < B; ++i) if (ids[i] < 0 || ids[i] >= K) return fail(PNP_E_ARG, "mask_id[%d]=%d out of range [0,%d)", i, ids[i], K);
        rc = copy_in(c, c->mask_id, mask_id, (size_t)B * sizeof(int32_t), on_device);
        if (rc) return rc;
    } else {
        HIPCHK(hipMemsetAsync(c->mask_id, 0, (size_t)B * sizeof(int32_t), c->stream));
    }
    c->B = B; c->K = K; c->have_x = false; c->fused_ready = false;
    return PNP_OK;
}

// One workgroup per slice, one workgroup per compute unit at a time: the batch runs in rounds of `cus` slices.
// Measured on MI355X, ms per iteration, two-launch vs slice-resident (profiles/run_slice_sizes.sh, one box each):
//   round 3 (profiles/slice_sizes_r03.txt):  B = 16: 0.0218 / 0.0289   32: 0.0253 / 0.0291   48: 0.0269 / 0.0302   56: 0.0290 / 0.0310
//                                            64: 0.0340 / 0.0309   96: 0.0437 / 0.0319
//   round 2:  B = 128: 0.0544 / 0.0405   256: 0.0954 / 0.0565   272: 0.1063 / 0.0904   320: 0.1224 / 0.0936   512: 0.1919 / 0.1075
// A round costs the same full or not, and even a nearly empty second round (B = 272) beats the two-launch path:
// the rule is simply "at least PNP_SLICE_MIN_B (64) slices" -- the crossover still sits between 56 and 64 with round 3's kernel.
static bool slice_pays(pnp_ctx* c) {
    return c->slice_force || c->B >= c->slice_min_b;
}

static int prepare_fused_tables(pnp_ctx* c);
// A failed prepare leaves no valid problem behind: B = 0, no table marked ready.
static int prepare_fused(pnp_ctx* c) {
    c->slice_ready = false;
    c->slice_tabs = false;
    c->fused_ready = false;
    const int rc = prepare_fused_tables(c);
    if (rc) { c->B = 0; c->slice_ready = false; c->fused_ready = false; }
    return rc;
}
// the two-launch tables of a 256x256 float context: needed by pnp_dc_step and by loops that do not take the slice-resident path
static int ensure_fused_tabs(pnp_ctx* c) {
    if (c->fused && !c->fused_tabs) {
        HIPCHK(fused256_prepare(c->fused, c->stream, c->y, c->mask_bank, c->mask_id, c->B));
        c->fused_tabs = true;
    }
    return PNP_OK;
}
// the slice-resident tables (256 KiB per slice, two launches): built when the first loop takes that path, so that the step-wise
// PnP solvers -- which only ever call pnp_dc_step on such a context -- never pay for them
static int ensure_slice_tabs(pnp_ctx* c) {
    if (c->slice && c->slice_ready && !c->slice_tabs) {
        HIPCHK(slice256_prepare(c->slice, c->stream, c->y, c->mask_bank, c->mask_id, c->B));
        c->slice_tabs = true;
    }
    return PNP_OK;
}
static int prepare_fused_tables(pnp_ctx* c) {
    if (c->fused) {
        c->fused_tabs = false;
        c->slice_tabs = false;
        if (c->slice && slice_pays(c)) {
            c->slice_ready = true;                        // tables: ensure_slice_tabs / ensure_fused_tabs, on first use
        } else {
            const int rc = ensure_fused_tabs(c);
            if (rc) return rc;
        }
        c->fused_ready = true;
    } else if (c->fused5) {
        HIPCHK(fused512_prepare(c->fused5, c->stream, c->y, c->mask_bank, c->mask_id, c->B));
        c->fused_ready = true;
    } else if (c->fs32) {
        HIPCHK(fused256s_prepare<float>(c->fs32, c->stream, c->y, c->mask_bank, c->mask_id, c->B));
        c->fused_ready = true;
    } else if (c->fs64) {
        HIPCHK(fused256s_prepare<double>(c->fs64, c->stream, c->yd, c->mask_bank, c->mask_id, c->B));
        c->fused_ready = true;
    }
    return PNP_OK;
}

int pnp_upload_problem(pnp_ctx* c, const float* y, const uint8_t* mask_bank, const int32_t* mask_id, int B, int K, int on_device) {
    CTX(c); F32_ONLY(c);
    if (!y) return fail(PNP_E_ARG, "pnp_upload_problem: y is null");
    // A new problem invalidates the state (include/pnp_mri.h): z / w are UNDEFINED until pnp_init_state or pnp_set_state(z, w).  The
    // order flag is reset with it -- a conversion under the new B would read the slice path's padded arrays beyond the old batch.
    c->state_sliced = false;
    c->have_x = false;
    c->have_state = false;
    int rc = set_masks(c, mask_bank, mask_id, B, K, on_device);
    if (rc) { c->B = 0; return rc; }
    rc = copy_in(c, c->y, y, (size_t)B * c->N * sizeof(float2), on_device);
    if (rc) { c->B = 0; return rc; }
    return prepare_fused(c);
}

int pnp_synthesize_problem(pnp_ctx* c, const float* img, const float* noise, int noise_per_slice,
                           const uint8_t* mask_bank, const int32_t* mask_id, int B, int K, int on_device) {
    CTX(c); F32_ONLY(c);
    if (!img || !noise) return fail(PNP_E_ARG, "pnp_synthesize_problem: img/noise is null");
    // A new problem invalidates the state (include/pnp_mri.h): z / w are UNDEFINED until pnp_init_state or pnp_set_state(z, w).  The
    // order flag is reset with it -- a conversion under the new B would read the slice path's padded arrays beyond the old batch.
    c->state_sliced = false;
    c->have_x = false;
    c->have_state = false;
    int rc = set_masks(c, mask_bank, mask_id, B, K, on_device);
    if (rc) { c->B = 0; return rc; }
    const size_t img_bytes = (size_t)B * c->N * sizeof(float);
    const size_t noise_bytes = (noise_per_slice ? (size_t)B : 1) * c->N * sizeof(float2);
    const float* d_img = img;
    const float2* d_noise = (const float2*)noise;
    if (!on_device) {
        const size_t need = img_bytes + noise_bytes;
        if (need > c->stage_bytes) {
            c->B = 0;                                   // no valid problem until y has been rebuilt
            if (c->stage) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->stage)); c->stage = nullptr; c->stage_bytes = 0; }
            HIPCHK(hipMalloc(&c->stage, need));
            c->B = B;
            c->stage_bytes = need;
        }
        rc = copy_in(c, c->stage, img, img_bytes, 0); if (rc) { c->B = 0; return rc; }
        rc = copy_in(c, (char*)c->stage + img_bytes, noise, noise_bytes, 0); if (rc) { c->B = 0; return rc; }
        d_img = (const float*)c->stage;
        d_noise = (const float2*)((char*)c->stage + img_bytes);
    }
    RowArgs ra{};
    ra.rin0 = d_img; ra.cout = c->y; ra.scale = 1.0f; ra.nrows = B * c->H;
    HIPCHK(launch_rows(c->stream, c->W, IN_REAL, false, EPI_COMPLEX, ra));
    ColArgs ca{};
    ca.in = c->y; ca.out = c->y; ca.y = d_noise; ca.mask_bank = c->mask_bank; ca.mask_id = c->mask_id;
    ca.y_per_slice = noise_per_slice; ca.B = B;
    HIPCHK(launch_cols(c->stream, c->H, c->W, true, MID_MASK_ADD, false, ca));
    return prepare_fused(c);
}

int pnp_download_y(pnp_ctx* c, float* y, int on_device) {
    CTX(c); F32_ONLY(c); NEED_PROBLEM(c);
    if (!y) return fail(PNP_E_ARG, "pnp_download_y: null");
    return copy_out(c, y, c->y, (size_t)c->B * c->N * sizeof(float2), on_device);
}

int pnp_init_state(pnp_ctx* c) {
    CTX(c); NEED_PROBLEM(c);
    if (c->f64) {
        ColArgsT<double> ca{};
        ca.in = c->yd; ca.out = c->workd; ca.B = c->B;
        HIPCHK(launch_cols<double>(c->stream, c->H, c->W, false, MID_NONE, true, ca));
        RowArgsT<double> ra{};
        ra.cin = c->workd; ra.x_out = c->zd; ra.scale = 1.0 / (double)c->N; ra.nrows = c->B * c->H;
        HIPCHK(launch_rows<double>(c->stream, c->W, IN_COMPLEX, true, EPI_ABS_COMPLEX, ra));
        HIPCHK(hipMemsetAsync(c->wd, 0, (size_t)c->B * c->N * sizeof(double), c->stream));
        c->have_x = false;
        c->have_state = true;
        return PNP_OK;
    }
    c->state_sliced = false;                           // both arrays are rewritten below, in natural order
    ColArgs ca{};
    ca.in = c->y; ca.out = c->work; ca.B = c->B;
    HIPCHK(launch_cols(c->stream, c->H, c->W, false, MID_NONE, true, ca));
    RowArgs ra{};
    ra.cin = c->work; ra.x_out = c->z; ra.scale = 1.0f / (float)c->N; ra.nrows = c->B * c->H;
    HIPCHK(launch_rows(c->stream, c->W, IN_COMPLEX, true, EPI_ABS_COMPLEX, ra));
    HIPCHK(hipMemsetAsync(c->w, 0, (size_t)c->B * c->N * sizeof(float), c->stream));
    c->have_x = false;
    c->have_state = true;
    return PNP_OK;
}

int pnp_set_state(pnp_ctx* c, const float* z, const float* w, int on_device) {
    CTX(c); F32_ONLY(c); NEED_PROBLEM(c);
    const size_t bytes = (size_t)c->B * c->N * sizeof(float);
    int rc;
    if (z && w) c->state_sliced = false;               // both replaced: nothing to convert
    else { NEED_STATE(c); if ((rc = state_order(c, false))) return rc; }      // one of the two kept: it must be defined
    if (z) { rc = copy_in(c, c->z, z, bytes, on_device); if (rc) return rc; }
    if (w) { rc = copy_in(c, c->w, w, bytes, on_device); if (rc) return rc; }
    c->have_x = false;
    if (z && w) c->have_state = true;
    return PNP_OK;
}

int pnp_get_state(pnp_ctx* c, float* z, float* w, int on_device) {
    CTX(c); F32_ONLY(c); NEED_PROBLEM(c); NEED_STATE(c);
    const size_t bytes = (size_t)c->B * c->N * sizeof(float);
    int rc;
    if ((rc = state_order(c, false))) return rc;
    if (z) { rc = copy_out(c, z, c->z, bytes, on_device); if (rc) return rc; }
    if (w) { rc = copy_out(c, w, c->w, bytes, on_device); if (rc) return rc; }
    return PNP_OK;
}

// one generic iteration: rows fwd (z-w) -> cols fwd/blend/inv -> rows inv + prox + dual
static int generic_iteration(pnp_ctx* c, const float* z_in, const float* w_in, RowEpi epi, const ProxParams& pp,
                             float cdc, float* x_out, float* z_io, float* w_io) {
    RowArgs ra{};
    ra.rin0 = z_in; ra.rin1 = w_in; ra.cout = c->work; ra.scale = 1.0f; ra.nrows = c->B * c->H;
    HIPCHK(launch_rows(c->stream, c->W, IN_REAL_DIFF, false, EPI_COMPLEX, ra));
    ColArgs ca{};
    ca.in = c->work; ca.out = c->work; ca.y = c->y; ca.mask_bank = c->mask_bank; ca.mask_id = c->mask_id;
    ca.c = cdc; ca.B = c->B;
    HIPCHK(launch_cols(c->stream, c->H, c->W, true, MID_BLEND, true, ca));
    RowArgs rb{};
    rb.cin = c->work; rb.x_out = x_out; rb.z = z_io; rb.w = w_io; rb.scale = 1.0f / (float)c->N;
    rb.prox = pp; rb.nrows = c->B * c->H;
    HIPCHK(launch_rows(c->stream, c->W, IN_COMPLEX, true, epi, rb));
    return PNP_OK;
}

static int run_loop_f64(pnp_ctx* c, int iters, bool cnc, const ProxParamsT<double>& pp, double reo) {
    const double cdc = 1.0 / (1.0 + 1.0 / 2.0 / reo);
    if (iters == 0) HIPCHK(hipMemcpyAsync(c->xd, c->zd, (size_t)c->B * c->N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    if (iters > 0 && use_fused(c)) {
        HIPCHK(fused256s_run<double>(c->fs64, c->stream, c->zd, c->wd, c->xd, c->B, iters, cnc, cdc, pp, c->sched));
        c->have_x = true;
        return PNP_OK;
    }
    for (int i = 0; i < iters; ++i) {
        RowArgsT<double> ra{};
        ra.rin0 = c->zd; ra.rin1 = c->wd; ra.cout = c->workd; ra.scale = 1.0; ra.nrows = c->B * c->H;
        HIPCHK(launch_rows<double>(c->stream, c->W, IN_REAL_DIFF, false, EPI_COMPLEX, ra));
        ColArgsT<double> ca{};
        ca.in = c->workd; ca.out = c->workd; ca.y = c->yd; ca.mask_bank = c->mask_bank; ca.mask_id = c->mask_id;
        ca.c = cdc; ca.B = c->B;
        HIPCHK(launch_cols<double>(c->stream, c->H, c->W, true, MID_BLEND, true, ca));
        RowArgsT<double> rb{};
        rb.cin = c->workd; rb.x_out = (i == iters - 1) ? c->xd : nullptr; rb.z = c->zd; rb.w = c->wd;
        rb.scale = 1.0 / (double)c->N; rb.prox = pp; rb.nrows = c->B * c->H;
        HIPCHK(launch_rows<double>(c->stream, c->W, IN_COMPLEX, true, cnc ? EPI_CNC : EPI_L1, rb));
    }
    c->have_x = true;
    return PNP_OK;
}

static int run_loop(pnp_ctx* c, int iters, bool cnc, const ProxParams& pp, double reo) {
    if (iters < 0) return fail(PNP_E_ARG, "iters must be >= 0");
    if (!(reo > 0.0)) return fail(PNP_E_ARG, "reo must be > 0");
    const bool slice_loop = iters > 0 && use_fused(c) && c->slice && c->slice_ready;
    if (slice_loop) { if (int rt = ensure_slice_tabs(c)) return rt; }
    if (int rs = state_order(c, slice_loop)) return rs;
    if (iters == 0) {
        // the reference's loop body never runs and its x stays the initial x = |ifft2(y)| = z0 (S4:103, 107, 138)
        HIPCHK(hipMemcpyAsync(c->x, c->z, (size_t)c->B * c->N * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        c->have_x = true;
        return PNP_OK;
    }
    const float cdc = dc_coeff(reo);
    if (use_fused(c)) {
        if (c->slice && c->slice_ready) HIPCHK(slice256_run(c->slice, c->stream, c->z, c->w, c->x, c->B, iters, cnc, cdc, pp, c->sched));
        else if (c->fused)  { const int rt = ensure_fused_tabs(c); if (rt) return rt; HIPCHK(fused256_run(c->fused, c->stream, c->z, c->w, c->x, c->B, iters, cnc, cdc, pp, c->sched)); }
        else if (c->fs32)   HIPCHK(fused256s_run<float>(c->fs32, c->stream, c->z, c->w, c->x, c->B, iters, cnc, cdc, pp, c->sched));
        else                HIPCHK(fused512_run(c->fused5, c->stream, c->z, c->w, c->x, c->B, iters, cnc, cdc, pp, c->sched));
    } else {
        for (int i = 0; i < iters; ++i) {
            int rc = generic_iteration(c, c->z, c->w, cnc ? EPI_CNC : EPI_L1, pp, cdc,
                                       (i == iters - 1) ? c->x : nullptr, c->z, c->w);
            if (rc) return rc;
        }
    }
    c->have_x = true;
    return PNP_OK;
}

// soft(a, thr) is evaluated as a - med3(a, -thr, thr) on the fast paths, which equals the reference's
// fmax(|a| - thr, 0) * sign(a) (S1:18-19) for thr >= 0 only; the reference's own parameters are all positive.
static int check_thresholds(const char* who, double alpha, double lambda1, double reo) {
    if (!(lambda1 >= 0.0)) return fail(PNP_E_ARG, "%s: lambda1 must be >= 0 (got %g)", who, lambda1);
    if (!(alpha >= 0.0)) return fail(PNP_E_ARG, "%s: alpha must be >= 0 (got %g)", who, alpha);
    if (!(reo > 0.0)) return fail(PNP_E_ARG, "%s: reo must be > 0 (got %g)", who, reo);
    return PNP_OK;
}

int pnp_admm_l1_run(pnp_ctx* c, int iters, double lambda1, double reo) {
    CTX(c); NEED_PROBLEM(c); NEED_STATE(c);
    Range r("pnp_admm_l1_run");
    if (int rv = check_thresholds("pnp_admm_l1_run", 0.0, lambda1, reo)) return rv;
    if (c->f64) {
        if (iters < 0 || !(reo > 0.0)) return fail(PNP_E_ARG, "pnp_admm_l1_run: iters >= 0 and reo > 0 required");
        ProxParamsT<double> p{}; p.thr = reo * lambda1;
        return run_loop_f64(c, iters, false, p, reo);
    }
    return run_loop(c, iters, false, make_prox_l1(lambda1, reo), reo);
}

int pnp_admm_cnc_run(pnp_ctx* c, int iters, double alpha, double lambda1, double reo, double b) {
    CTX(c); NEED_PROBLEM(c); NEED_STATE(c);
    Range r("pnp_admm_cnc_run");
    if (!(b > 0.0)) return fail(PNP_E_ARG, "pnp_admm_cnc_run: b must be > 0");
    if (int rv = check_thresholds("pnp_admm_cnc_run", alpha, lambda1, reo)) return rv;
    if (c->f64) {
        if (iters < 0 || !(reo > 0.0)) return fail(PNP_E_ARG, "pnp_admm_cnc_run: iters >= 0 and reo > 0 required");
        ProxParamsT<double> p{};
        p.thr = alpha * reo * lambda1; p.c1 = 1.0 - alpha; p.c2 = alpha; p.c3 = alpha * reo * lambda1 * b; p.ib = 1.0 / b;
        return run_loop_f64(c, iters, true, p, reo);
    }
    return run_loop(c, iters, true, make_prox_cnc(alpha, lambda1, reo, b), reo);
}

int pnp_download_x(pnp_ctx* c, float* x, int on_device) {
    CTX(c); F32_ONLY(c); NEED_PROBLEM(c);
    if (!x) return fail(PNP_E_ARG, "pnp_download_x: null");
    if (!c->have_x) return fail(PNP_E_STATE, "pnp_download_x: no iteration has been run since the state was set");
    return copy_out(c, x, c->x, (size_t)c->B * c->N * sizeof(float), on_device);
}

int pnp_dc_step(pnp_ctx* c, const float* z, const float* w, float* x, double reo) {
    CTX(c); F32_ONLY(c); NEED_PROBLEM(c);
    Range r("pnp_dc_step");
    if (!z || !w || !x) return fail(PNP_E_ARG, "pnp_dc_step: null pointer");
    if (!(reo > 0.0)) return fail(PNP_E_ARG, "pnp_dc_step: reo must be > 0");
    if (use_fused(c)) {
        if (c->fused)     { const int rt = ensure_fused_tabs(c); if (rt) return rt; HIPCHK(fused256_dc(c->fused, c->stream, z, w, x, c->B, dc_coeff(reo))); }
        else if (c->fs32) HIPCHK(fused256s_dc<float>(c->fs32, c->stream, z, w, x, c->B, dc_coeff(reo)));
        else              HIPCHK(fused512_dc(c->fused5, c->stream, z, w, x, c->B, dc_coeff(reo)));
        return PNP_OK;
    }
    return generic_iteration(c, z, w, EPI_ABS_REAL, ProxParams{}, dc_coeff(reo), x, nullptr, nullptr);
}

int pnp_prox_l1_dual(pnp_ctx* c, const float* x, float* z, float* w, double thr) {
    CTX(c); F32_ONLY(c); NEED_PROBLEM(c);
    if (!x || !z || !w) return fail(PNP_E_ARG, "pnp_prox_l1_dual: null pointer");
    if (!(thr >= 0.0)) return fail(PNP_E_ARG, "pnp_prox_l1_dual: thr must be >= 0");
    ProxParams p{}; p.thr = (float)thr;
    HIPCHK(launch_prox(c->stream, false, x, z, w, p, (size_t)c->B * c->N));
    return PNP_OK;
}

int pnp_prox_cnc_dual(pnp_ctx* c, const float* x, float* z, float* w, double alpha, double lambda1, double reo, double b) {
    CTX(c); F32_ONLY(c); NEED_PROBLEM(c);
    if (!x || !z || !w) return fail(PNP_E_ARG, "pnp_prox_cnc_dual: null pointer");
    if (!(b > 0.0)) return fail(PNP_E_ARG, "pnp_prox_cnc_dual: b must be > 0");
    if (int rv = check_thresholds("pnp_prox_cnc_dual", alpha, lambda1, reo)) return rv;
    HIPCHK(launch_prox(c->stream, true, x, z, w, make_prox_cnc(alpha, lambda1, reo, b), (size_t)c->B * c->N));
    return PNP_OK;
}

int pnp_cnc_combine(pnp_ctx* c, const float* z, const float* x, const float* w, const float* s, float* t,
                    double alpha, double lambda1, double reo, double b) {
    CTX(c); F32_ONLY(c); NEED_PROBLEM(c);
    if (!z || !x || !w || !s || !t) return fail(PNP_E_ARG, "pnp_cnc_combine: null pointer");
    HIPCHK(launch_combine(c->stream, z, x, w, s, t, (float)(1.0 - alpha), (float)alpha,
                          (float)(alpha * reo * lambda1 * b), (size_t)c->B * c->N));
    return PNP_OK;
}

int pnp_add(pnp_ctx* c, const float* a, const float* b, float* o) {
    CTX(c); F32_ONLY(c); NEED_PROBLEM(c);
    if (!a || !b || !o) return fail(PNP_E_ARG, "pnp_add: null pointer");
    HIPCHK(launch_add(c->stream, a, b, o, (size_t)c->B * c->N));
    return PNP_OK;
}

int pnp_dual_clamp(pnp_ctx* c, float* x, float* z, float* w) {
    CTX(c); F32_ONLY(c); NEED_PROBLEM(c);
    if (!x || !z || !w) return fail(PNP_E_ARG, "pnp_dual_clamp: null pointer");
    HIPCHK(launch_dual_clamp(c->stream, x, z, w, (size_t)c->B * c->N));
    return PNP_OK;
}

static int fft2_any(pnp_ctx* c, const float* in, float* out, int B, bool inv) {
    if (!in || !out) return fail(PNP_E_ARG, "fft2: null pointer");
    if (B < 1 || B > c->Bmax) return fail(PNP_E_ARG, "fft2: B=%d out of range [1,%d]", B, c->Bmax);
    RowArgs ra{};
    ra.cin = (const float2*)in; ra.cout = (float2*)out; ra.scale = inv ? 1.0f / (float)c->N : 1.0f; ra.nrows = B * c->H;
    HIPCHK(launch_rows(c->stream, c->W, IN_COMPLEX, inv, EPI_COMPLEX, ra));
    ColArgs ca{};
    ca.in = (const float2*)out; ca.out = (float2*)out; ca.B = B;
    HIPCHK(launch_cols(c->stream, c->H, c->W, !inv, MID_NONE, inv, ca));
    return PNP_OK;
}

int pnp_fft2_fwd(pnp_ctx* c, const float* in, float* out, int B) { CTX(c); F32_ONLY(c); return fft2_any(c, in, out, B, false); }
int pnp_fft2_inv(pnp_ctx* c, const float* in, float* out, int B) { CTX(c); F32_ONLY(c); return fft2_any(c, in, out, B, true); }

int pnp_A(pnp_ctx* c, const float* x, float* k) {
    CTX(c); F32_ONLY(c); NEED_PROBLEM(c);
    if (!x || !k) return fail(PNP_E_ARG, "pnp_A: null pointer");
    RowArgs ra{};
    ra.rin0 = x; ra.cout = (float2*)k; ra.scale = 1.0f; ra.nrows = c->B * c->H;
    HIPCHK(launch_rows(c->stream, c->W, IN_REAL, false, EPI_COMPLEX, ra));
    ColArgs ca{};
    ca.in = (const float2*)k; ca.out = (float2*)k; ca.mask_bank = c->mask_bank; ca.mask_id = c->mask_id; ca.B = c->B;
    HIPCHK(launch_cols(c->stream, c->H, c->W, true, MID_MASK, false, ca));
    return PNP_OK;
}

int pnp_AH(pnp_ctx* c, const float* k, float* out) {
    CTX(c); F32_ONLY(c); NEED_PROBLEM(c);
    if (!k || !out) return fail(PNP_E_ARG, "pnp_AH: null pointer");
    ColArgs ca{};
    ca.in = (const float2*)k; ca.out = (float2*)out; ca.mask_bank = c->mask_bank; ca.mask_id = c->mask_id; ca.B = c->B;
    HIPCHK(launch_cols(c->stream, c->H, c->W, false, MID_MASK, true, ca));
    RowArgs ra{};
    ra.cin = (const float2*)out; ra.cout = (float2*)out; ra.scale = 1.0f / (float)c->N; ra.nrows = c->B * c->H;
    HIPCHK(launch_rows(c->stream, c->W, IN_COMPLEX, true, EPI_COMPLEX, ra));
    return PNP_OK;
}

int pnp_Df(pnp_ctx* c, const float* x, float* out) {
    CTX(c); F32_ONLY(c); NEED_PROBLEM(c);
    if (!x || !out) return fail(PNP_E_ARG, "pnp_Df: null pointer");
    RowArgs ra{};
    ra.rin0 = x; ra.cout = (float2*)out; ra.scale = 1.0f; ra.nrows = c->B * c->H;
    HIPCHK(launch_rows(c->stream, c->W, IN_REAL, false, EPI_COMPLEX, ra));
    ColArgs ca{};
    ca.in = (const float2*)out; ca.out = (float2*)out; ca.y = c->y; ca.mask_bank = c->mask_bank; ca.mask_id = c->mask_id; ca.B = c->B;
    HIPCHK(launch_cols(c->stream, c->H, c->W, true, MID_RESID, true, ca));
    RowArgs rb{};
    rb.cin = (const float2*)out; rb.cout = (float2*)out; rb.scale = 1.0f / (float)c->N; rb.nrows = c->B * c->H;
    HIPCHK(launch_rows(c->stream, c->W, IN_COMPLEX, true, EPI_COMPLEX, rb));
    return PNP_OK;
}

}  // extern "C"  (templates have C++ linkage)

template <typename X>
static int metrics_any(pnp_ctx* c, const X* x, const X* own_x, const uint8_t* gt, int gt_on_device, double* psnr, double* re) {
    if (!gt || !psnr || !re) return fail(PNP_E_ARG, "pnp_metrics: null pointer");
    if (!x) {
        if (!c->have_x) return fail(PNP_E_STATE, "pnp_metrics: x_dev is null and the ctx holds no x yet");
        x = own_x;
    }
    const uint8_t* d_gt = gt;
    if (!gt_on_device) {
        if (!c->gt) HIPCHK(hipMalloc((void**)&c->gt, (size_t)c->Bmax * c->N));
        int rc = copy_in(c, c->gt, gt, (size_t)c->B * c->N, 0);
        if (rc) return rc;
        d_gt = c->gt;
    }
    HIPCHK(launch_metrics<X>(c->stream, x, d_gt, c->acc, c->B, (int)c->N));
    std::vector<double> h((size_t)c->B * 2);
    int rc = copy_out(c, h.data(), c->acc, h.size() * sizeof(double), 0);
    if (rc) return rc;
    for (int b = 0; b < c->B; ++b) {
        const double mse = h[2 * b] / (double)c->N;
        psnr[b] = (mse == 0.0) ? INFINITY : 20.0 * log10(255.0 / sqrt(mse));
        re[b] = sqrt(h[2 * b]) / sqrt(h[2 * b + 1]);
    }
    return PNP_OK;
}

template <typename X>
static int ssim_any(pnp_ctx* c, const X* x, const X* own_x, const uint8_t* gt, int gt_on_device, double* ssim) {
    if (!gt || !ssim) return fail(PNP_E_ARG, "pnp_ssim: null pointer");
    if (!x) {
        if (!c->have_x) return fail(PNP_E_STATE, "pnp_ssim: x_dev is null and the ctx holds no x yet");
        x = own_x;
    }
    const uint8_t* d_gt = gt;
    if (!gt_on_device) {
        if (!c->gt) HIPCHK(hipMalloc((void**)&c->gt, (size_t)c->Bmax * c->N));
        int rc = copy_in(c, c->gt, gt, (size_t)c->B * c->N, 0);
        if (rc) return rc;
        d_gt = c->gt;
    }
    const int tiles = ((c->W - 10 + 15) / 16) * ((c->H - 10 + 15) / 16);
    if (!c->ssim_part) HIPCHK(hipMalloc((void**)&c->ssim_part, (size_t)c->Bmax * tiles * sizeof(double)));
    HIPCHK(launch_ssim<X>(c->stream, x, d_gt, c->ssim_part, c->B, c->H, c->W));
    std::vector<double> h((size_t)c->B * tiles);
    int rc = copy_out(c, h.data(), c->ssim_part, h.size() * sizeof(double), 0);
    if (rc) return rc;
    const double npix = (double)(c->H - 10) * (double)(c->W - 10);
    for (int b = 0; b < c->B; ++b) {
        double s = 0.0;
        for (int t = 0; t < tiles; ++t) s += h[(size_t)b * tiles + t];
        ssim[b] = s / npix;
    }
    return PNP_OK;
}

extern "C" {

int pnp_metrics(pnp_ctx* c, const float* x, const uint8_t* gt, int gt_on_device, double* psnr, double* re) {
    CTX(c); F32_ONLY(c); NEED_PROBLEM(c);
    return metrics_any<float>(c, x, c->x, gt, gt_on_device, psnr, re);
}
int pnp_ssim(pnp_ctx* c, const float* x, const uint8_t* gt, int gt_on_device, double* ssim) {
    CTX(c); F32_ONLY(c); NEED_PROBLEM(c);
    return ssim_any<float>(c, x, c->x, gt, gt_on_device, ssim);
}
int pnp_metrics_f64(pnp_ctx* c, const double* x, const uint8_t* gt, int gt_on_device, double* psnr, double* re) {
    CTX(c); F64_ONLY(c); NEED_PROBLEM(c);
    return metrics_any<double>(c, x, c->xd, gt, gt_on_device, psnr, re);
}
int pnp_ssim_f64(pnp_ctx* c, const double* x, const uint8_t* gt, int gt_on_device, double* ssim) {
    CTX(c); F64_ONLY(c); NEED_PROBLEM(c);
    return ssim_any<double>(c, x, c->xd, gt, gt_on_device, ssim);
}

/* ---- fp64 validation context: problem / state / result in double ------------------------- */
int pnp_upload_problem_f64(pnp_ctx* c, const double* y, const uint8_t* mask_bank, const int32_t* mask_id, int B, int K, int on_device) {
    CTX(c); F64_ONLY(c);
    if (!y) return fail(PNP_E_ARG, "pnp_upload_problem_f64: y is null");
    c->have_x = false; c->have_state = false;           // a new problem: z / w undefined until pnp_init_state / pnp_set_state_f64(z, w)
    int rc = set_masks(c, mask_bank, mask_id, B, K, on_device);
    if (rc) { c->B = 0; return rc; }
    rc = copy_in(c, c->yd, y, (size_t)B * c->N * sizeof(double2), on_device);
    if (rc) { c->B = 0; return rc; }
    return prepare_fused(c);
}

// y = fft2(img) * mask + noise in double (S4:102).  The reference's first fft2 runs on the float32 image in complex64
// (NumPy >= 2) and is promoted by the float64 mask; here the float32 image is widened (exactly) and transformed in double,
// which is the nearer of the two to the exact transform -- the two y differ by NumPy's own complex64 round-off, ~1e-7.
int pnp_synthesize_problem_f64(pnp_ctx* c, const float* img, const double* noise, int noise_per_slice,
                               const uint8_t* mask_bank, const int32_t* mask_id, int B, int K, int on_device) {
    CTX(c); F64_ONLY(c);
    if (!img || !noise) return fail(PNP_E_ARG, "pnp_synthesize_problem_f64: img/noise is null");
    c->have_x = false; c->have_state = false;
    int rc = set_masks(c, mask_bank, mask_id, B, K, on_device);
    if (rc) { c->B = 0; return rc; }
    const size_t img_bytes = (size_t)B * c->N * sizeof(float);
    const size_t noise_bytes = (noise_per_slice ? (size_t)B : 1) * c->N * sizeof(double2);
    const float* d_img = img;
    const double2* d_noise = (const double2*)noise;
    if (!on_device) {
        const size_t need = img_bytes + noise_bytes;
        if (need > c->stage_bytes) {
            c->B = 0;                                   // no valid problem until y has been rebuilt
            if (c->stage) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->stage)); c->stage = nullptr; c->stage_bytes = 0; }
            HIPCHK(hipMalloc(&c->stage, need));
            c->B = B;
            c->stage_bytes = need;
        }
        rc = copy_in(c, (char*)c->stage, noise, noise_bytes, 0); if (rc) { c->B = 0; return rc; }      // doubles first: alignment
        rc = copy_in(c, (char*)c->stage + noise_bytes, img, img_bytes, 0); if (rc) { c->B = 0; return rc; }
        d_noise = (const double2*)c->stage;
        d_img = (const float*)((char*)c->stage + noise_bytes);
    }
    HIPCHK(launch_widen(c->stream, d_img, c->xd, (size_t)B * c->N));     // x is invalid until the next run anyway (have_x = false)
    RowArgsT<double> ra{};
    ra.rin0 = c->xd; ra.cout = c->yd; ra.scale = 1.0; ra.nrows = B * c->H;
    HIPCHK(launch_rows<double>(c->stream, c->W, IN_REAL, false, EPI_COMPLEX, ra));
    ColArgsT<double> ca{};
    ca.in = c->yd; ca.out = c->yd; ca.y = d_noise; ca.mask_bank = c->mask_bank; ca.mask_id = c->mask_id;
    ca.y_per_slice = noise_per_slice; ca.B = B;
    HIPCHK(launch_cols<double>(c->stream, c->H, c->W, true, MID_MASK_ADD, false, ca));
    return prepare_fused(c);
}

int pnp_download_y_f64(pnp_ctx* c, double* y, int on_device) {
    CTX(c); F64_ONLY(c); NEED_PROBLEM(c);
    if (!y) return fail(PNP_E_ARG, "pnp_download_y_f64: null");
    return copy_out(c, y, c->yd, (size_t)c->B * c->N * sizeof(double2), on_device);
}

int pnp_set_state_f64(pnp_ctx* c, const double* z, const double* w, int on_device) {
    CTX(c); F64_ONLY(c); NEED_PROBLEM(c);
    const size_t bytes = (size_t)c->B * c->N * sizeof(double);
    int rc;
    if (!(z && w)) NEED_STATE(c);                          // one of the two kept: it must be defined
    if (z) { rc = copy_in(c, c->zd, z, bytes, on_device); if (rc) return rc; }
    if (w) { rc = copy_in(c, c->wd, w, bytes, on_device); if (rc) return rc; }
    c->have_x = false;
    if (z && w) c->have_state = true;
    return PNP_OK;
}

int pnp_get_state_f64(pnp_ctx* c, double* z, double* w, int on_device) {
    CTX(c); F64_ONLY(c); NEED_PROBLEM(c); NEED_STATE(c);
    const size_t bytes = (size_t)c->B * c->N * sizeof(double);
    int rc;
    if (z) { rc = copy_out(c, z, c->zd, bytes, on_device); if (rc) return rc; }
    if (w) { rc = copy_out(c, w, c->wd, bytes, on_device); if (rc) return rc; }
    return PNP_OK;
}

int pnp_download_x_f64(pnp_ctx* c, double* x, int on_device) {
    CTX(c); F64_ONLY(c); NEED_PROBLEM(c);
    if (!x) return fail(PNP_E_ARG, "pnp_download_x_f64: null");
    if (!c->have_x) return fail(PNP_E_STATE, "pnp_download_x_f64: no iteration has been run since the state was set");
    return copy_out(c, x, c->xd, (size_t)c->B * c->N * sizeof(double), on_device);
}

int pnp_is_f64(pnp_ctx* c) { return (c && c->f64) ? 1 : 0; }

/* ---- optional HIP backend of the denoisers' 64-channel body layers (no ctx: caller-owned device tensors) ---- */
int pnp_conv3x3_c64_nhwc(void* stream, const float* x, const float* w, const float* bias, const float* skip, float* y,
                         int n, int H, int W, int relu, int dilation) {
    if (!x || !w || !y) return fail(PNP_E_ARG, "pnp_conv3x3_c64_nhwc: null pointer");
    if (n < 1 || H < 1 || W < 1) return fail(PNP_E_ARG, "pnp_conv3x3_c64_nhwc: n, H, W must be >= 1");
    if (x == y || skip == y) return fail(PNP_E_ARG, "pnp_conv3x3_c64_nhwc: y must not alias x or skip (tiles read their neighbours' halo)");
    if (dilation < 1 || dilation > 4) return fail(PNP_E_ARG, "pnp_conv3x3_c64_nhwc: dilation must be 1..4 (got %d)", dilation);
    HIPCHK(launch_conv3x3_c64((hipStream_t)stream, x, w, bias, skip, y, n, H, W, relu, dilation));
    return PNP_OK;
}
int pnp_conv3x3_c64_pack(void* stream, const float* w_oihw, float* w_packed) {
    if (!w_oihw || !w_packed || w_oihw == w_packed) return fail(PNP_E_ARG, "pnp_conv3x3_c64_pack: null or aliased pointers");
    HIPCHK(launch_conv_pack_w((hipStream_t)stream, w_oihw, w_packed));
    return PNP_OK;
}
int pnp_conv3x3_c64_nhwc_f16x3(void* stream, const float* x, const float* w, const float* bias, const float* skip, float* y,
                               int n, int H, int W, int relu, int dilation) {
    if (!x || !w || !y) return fail(PNP_E_ARG, "pnp_conv3x3_c64_nhwc_f16x3: null pointer");
    if (n < 1 || H < 1 || W < 1) return fail(PNP_E_ARG, "pnp_conv3x3_c64_nhwc_f16x3: n, H, W must be >= 1");
    if (x == y || skip == y) return fail(PNP_E_ARG, "pnp_conv3x3_c64_nhwc_f16x3: y must not alias x or skip (tiles read their neighbours' halo)");
    if (dilation < 1 || dilation > 4) return fail(PNP_E_ARG, "pnp_conv3x3_c64_nhwc_f16x3: dilation must be 1..4 (got %d)", dilation);
    HIPCHK(launch_conv3x3_f16x3((hipStream_t)stream, x, w, bias, skip, y, n, 64, H, W, relu, dilation));
    return PNP_OK;
}
int pnp_conv3x3_nhwc_f16x3(void* stream, const float* x, const float* w, const float* bias, const float* skip, float* y,
                           int n, int C, int H, int W, int relu) {
    if (!x || !w || !y) return fail(PNP_E_ARG, "pnp_conv3x3_nhwc_f16x3: null pointer");
    if (n < 1 || H < 1 || W < 1) return fail(PNP_E_ARG, "pnp_conv3x3_nhwc_f16x3: n, H, W must be >= 1");
    if (C < 64 || C > 1024 || C % 64) return fail(PNP_E_ARG, "pnp_conv3x3_nhwc_f16x3: C must be a multiple of 64 in 64..1024 (got %d)", C);
    if ((long long)H * W * C * 4 > 0x7fffffffLL) return fail(PNP_E_ARG, "pnp_conv3x3_nhwc_f16x3: an image of %d x %d x %d floats exceeds 2 GiB", H, W, C);
    if (x == y || skip == y) return fail(PNP_E_ARG, "pnp_conv3x3_nhwc_f16x3: y must not alias x or skip (tiles read their neighbours' halo)");
    HIPCHK(launch_conv3x3_f16x3((hipStream_t)stream, x, w, bias, skip, y, n, C, H, W, relu, 1));
    return PNP_OK;
}
int pnp_conv3x3_nhwc_f16x3_fmt(void* stream, const float* x, const float* w, const float* bias, const float* skip, float* y,
                               int n, int C, int H, int W, int relu, int dilation, int fmt) {
    if (!x || !w || !y) return fail(PNP_E_ARG, "pnp_conv3x3_nhwc_f16x3_fmt: null pointer");
    if (n < 1 || H < 1 || W < 1) return fail(PNP_E_ARG, "pnp_conv3x3_nhwc_f16x3_fmt: n, H, W must be >= 1");
    if (C < 64 || C > 1024 || C % 64) return fail(PNP_E_ARG, "pnp_conv3x3_nhwc_f16x3_fmt: C must be a multiple of 64 in 64..1024 (got %d)", C);
    if (dilation < 1 || dilation > 4 || (C != 64 && dilation != 1)) return fail(PNP_E_ARG, "pnp_conv3x3_nhwc_f16x3_fmt: dilation 1..4 at C = 64, 1 otherwise (got %d)", dilation);
    if (fmt & ~7) return fail(PNP_E_ARG, "pnp_conv3x3_nhwc_f16x3_fmt: fmt is a mask of PNP_FMT_X_SPLIT | PNP_FMT_SKIP_SPLIT | PNP_FMT_Y_SPLIT (got %d)", fmt);
    if ((long long)H * W * C * 4 > 0x7fffffffLL) return fail(PNP_E_ARG, "pnp_conv3x3_nhwc_f16x3_fmt: an image of %d x %d x %d floats exceeds 2 GiB", H, W, C);
    if (x == y || skip == y) return fail(PNP_E_ARG, "pnp_conv3x3_nhwc_f16x3_fmt: y must not alias x or skip (tiles read their neighbours' halo)");
    HIPCHK(launch_conv3x3_f16x3((hipStream_t)stream, x, w, bias, skip, y, n, C, H, W, relu, dilation, fmt));
    return PNP_OK;
}
int pnp_conv3x3_f16x3_set_variant(int variant) { return conv_set_wide_mode(variant); }
int pnp_conv3x3_pack_f16x3(void* stream, const float* w_oihw, float* w_packed, int C) {
    if (!w_oihw || !w_packed || w_oihw == w_packed) return fail(PNP_E_ARG, "pnp_conv3x3_pack_f16x3: null or aliased pointers");
    if (C < 64 || C > 1024 || C % 64) return fail(PNP_E_ARG, "pnp_conv3x3_pack_f16x3: C must be a multiple of 64 in 64..1024 (got %d)", C);
    HIPCHK(launch_conv_pack_w_f16x3((hipStream_t)stream, w_oihw, w_packed, C));
    return PNP_OK;
}
int pnp_conv3x3_c64_pack_f16x3(void* stream, const float* w_oihw, float* w_packed) {
    if (!w_oihw || !w_packed || w_oihw == w_packed) return fail(PNP_E_ARG, "pnp_conv3x3_c64_pack_f16x3: null or aliased pointers");
    HIPCHK(launch_conv_pack_w_f16x3((hipStream_t)stream, w_oihw, w_packed, 64));
    return PNP_OK;
}
int pnp_conv3x3_head_nhwc(void* stream, const float* x, const float* w, const float* bias, float* y, int n, int cin, int H, int W, int relu) {
    if (!x || !w || !y) return fail(PNP_E_ARG, "pnp_conv3x3_head_nhwc: null pointer");
    if (n < 1 || H < 1 || W < 1 || cin < 1 || cin > 8) return fail(PNP_E_ARG, "pnp_conv3x3_head_nhwc: n, H, W >= 1 and 1 <= cin <= 8 required");
    HIPCHK(launch_conv3x3_head((hipStream_t)stream, x, w, bias, y, n, cin, H, W, relu));
    return PNP_OK;
}
int pnp_conv3x3_tail_nchw(void* stream, const float* x, const float* w, const float* bias, float* y, int n, int cout, int H, int W) {
    if (!x || !w || !y) return fail(PNP_E_ARG, "pnp_conv3x3_tail_nchw: null pointer");
    if (n < 1 || H < 1 || W < 1 || cout < 1 || cout > 4) return fail(PNP_E_ARG, "pnp_conv3x3_tail_nchw: n, H, W >= 1 and 1 <= cout <= 4 required");
    HIPCHK(launch_conv3x3_tail((hipStream_t)stream, x, w, bias, y, n, cout, H, W));
    return PNP_OK;
}
int pnp_conv3x3_tail_nchw_f16x3(void* stream, const float* x, const float* w, const float* bias, float* y, int n, int cout, int H, int W) {
    if (!x || !w || !y) return fail(PNP_E_ARG, "pnp_conv3x3_tail_nchw_f16x3: null pointer");
    if (n < 1 || H < 1 || W < 1 || cout < 1 || cout > 4) return fail(PNP_E_ARG, "pnp_conv3x3_tail_nchw_f16x3: n, H, W >= 1 and 1 <= cout <= 4 required");
    HIPCHK(launch_conv3x3_tail_f16x3((hipStream_t)stream, x, nullptr, w, bias, y, n, cout, H, W));
    return PNP_OK;
}
int pnp_conv3x3_tail_add_nchw_f16x3(void* stream, const float* x, const float* x2, const float* w, const float* bias, float* y, int n, int cout,
                                    int H, int W) {
    if (!x || !x2 || !w || !y) return fail(PNP_E_ARG, "pnp_conv3x3_tail_add_nchw_f16x3: null pointer");
    if (n < 1 || H < 1 || W < 1 || cout < 1 || cout > 4) return fail(PNP_E_ARG, "pnp_conv3x3_tail_add_nchw_f16x3: n, H, W >= 1 and 1 <= cout <= 4 required");
    if ((long long)H * W * 64 * 4 > 0x7fffffffLL) return fail(PNP_E_ARG, "pnp_conv3x3_tail_add_nchw_f16x3: an image of %d x %d x 64 floats exceeds 2 GiB", H, W);
    if (x == y || x2 == y) return fail(PNP_E_ARG, "pnp_conv3x3_tail_add_nchw_f16x3: y must not alias x or x2");
    HIPCHK(launch_conv3x3_tail_f16x3((hipStream_t)stream, x, x2, w, bias, y, n, cout, H, W));
    return PNP_OK;
}
int pnp_ffdnet_head_nhwc(void* stream, const float* x, const float* sigma, int sigma_per_image, const float* w, const float* bias, float* y,
                         int n, int h, int wd, int relu) {
    if (!x || !sigma || !w || !y) return fail(PNP_E_ARG, "pnp_ffdnet_head_nhwc: null pointer");
    if (n < 1 || h < 1 || wd < 1) return fail(PNP_E_ARG, "pnp_ffdnet_head_nhwc: n, h, w must be >= 1");
    if ((long long)((h + 1) / 2) * ((wd + 1) / 2) * 64 * 4 > 0x7fffffffLL) return fail(PNP_E_ARG, "pnp_ffdnet_head_nhwc: a result of %d x %d x 64 floats exceeds 2 GiB", (h + 1) / 2, (wd + 1) / 2);
    HIPCHK(launch_ffdnet_head((hipStream_t)stream, x, sigma, sigma_per_image != 0, w, bias, y, n, h, wd, relu));
    return PNP_OK;
}
int pnp_ffdnet_tail_f16x3(void* stream, const float* x, const float* w, const float* bias, float* y, int n, int h, int wd) {
    if (!x || !w || !y) return fail(PNP_E_ARG, "pnp_ffdnet_tail_f16x3: null pointer");
    if (n < 1 || h < 1 || wd < 1) return fail(PNP_E_ARG, "pnp_ffdnet_tail_f16x3: n, h, w must be >= 1");
    if ((long long)((h + 1) / 2) * ((wd + 1) / 2) * 64 * 4 > 0x7fffffffLL) return fail(PNP_E_ARG, "pnp_ffdnet_tail_f16x3: an input of %d x %d x 64 floats exceeds 2 GiB", (h + 1) / 2, (wd + 1) / 2);
    HIPCHK(launch_conv3x3_tail_f16x3((hipStream_t)stream, x, nullptr, w, bias, y, n, 4, (h + 1) / 2, (wd + 1) / 2, h, wd));
    return PNP_OK;
}
static int pix2_args(const char* who, const float* x, const float* x2, const float* w, const float* y, int n, int C, int H, int W, int up) {
    if (!x || !w || !y) return fail(PNP_E_ARG, "%s: null pointer", who);
    if (n < 1 || H < 1 || W < 1) return fail(PNP_E_ARG, "%s: n, H, W must be >= 1", who);
    if (C < 64 || C > 1024 || C % (up ? 128 : 64)) return fail(PNP_E_ARG, "%s: C must be a multiple of %d in 64..1024 (got %d)", who, up ? 128 : 64, C);
    if (!up && ((H | W) & 1)) return fail(PNP_E_ARG, "%s: H and W must be even (got %d x %d)", who, H, W);
    const long long in_b = (long long)H * W * C * 4, out_b = up ? in_b * 2 : in_b / 2;
    if (in_b > 0x7fffffffLL || out_b > 0x7fffffffLL) return fail(PNP_E_ARG, "%s: an image of %d x %d x %d floats (or its result) exceeds 2 GiB", who, H, W, C);
    if (x == y || x2 == y) return fail(PNP_E_ARG, "%s: y must not alias x or x2 (tiles are re-read after their neighbours were written)", who);
    return PNP_OK;
}
int pnp_conv2x2s2_nhwc_f16x3(void* stream, const float* x, const float* x2, const float* w, float* y, int n, int C, int H, int W) {
    if (int rc = pix2_args("pnp_conv2x2s2_nhwc_f16x3", x, x2, w, y, n, C, H, W, 0)) return rc;
    HIPCHK(launch_pix2x2_f16x3((hipStream_t)stream, x, x2, w, y, n, C, H, W, 0));
    return PNP_OK;
}
int pnp_convT2x2s2_nhwc_f16x3(void* stream, const float* x, const float* x2, const float* w, float* y, int n, int C, int H, int W) {
    if (int rc = pix2_args("pnp_convT2x2s2_nhwc_f16x3", x, x2, w, y, n, C, H, W, 1)) return rc;
    HIPCHK(launch_pix2x2_f16x3((hipStream_t)stream, x, x2, w, y, n, C, H, W, 1));
    return PNP_OK;
}
int pnp_conv2x2_pack_f16x3(void* stream, const float* w, float* w_packed, int C, int transposed) {
    if (!w || !w_packed || w == w_packed) return fail(PNP_E_ARG, "pnp_conv2x2_pack_f16x3: null or aliased pointers");
    if (C < 64 || C > 1024 || C % (transposed ? 128 : 64)) return fail(PNP_E_ARG, "pnp_conv2x2_pack_f16x3: C must be a multiple of %d in 64..1024 (got %d)", transposed ? 128 : 64, C);
    HIPCHK(launch_pix2_pack_w_f16x3((hipStream_t)stream, w, w_packed, C, transposed != 0));
    return PNP_OK;
}
int pnp_relayout_c64(void* stream, const float* in, float* out, int n, int H, int W, int to_nhwc) {
    if (!in || !out || in == out) return fail(PNP_E_ARG, "pnp_relayout_c64: null or aliased pointers");
    if (n < 1 || H < 1 || W < 1) return fail(PNP_E_ARG, "pnp_relayout_c64: n, H, W must be >= 1");
    HIPCHK(launch_relayout64((hipStream_t)stream, in, out, n, H * W, to_nhwc != 0));
    return PNP_OK;
}

int pnp_timer_start(pnp_ctx* c) { CTX(c); HIPCHK(hipEventRecord(c->ev0, c->stream)); return PNP_OK; }
int pnp_timer_stop(pnp_ctx* c, float* ms) {
    CTX(c);
    if (!ms) return fail(PNP_E_ARG, "pnp_timer_stop: null");
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev1));
    HIPCHK(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return PNP_OK;
}

// Build now whatever per-problem tables the loops (pnp_admm_*_run) of the uploaded problem will use; they are otherwise built
// by the first loop call (256x256 float contexts only -- every other path prepares at upload).  Benchmarks call it so that a
// timed region with no warm-up holds iterations only.
int pnp_prepare_loops(pnp_ctx* c) {
    CTX(c); NEED_PROBLEM(c);
    if (!c->fused || !use_fused(c)) return PNP_OK;
    if (c->slice && c->slice_ready) return ensure_slice_tabs(c);
    return ensure_fused_tabs(c);
}

int pnp_get_plan(pnp_ctx* c, int* queues, int* chunk, int* launches_per_iteration) {
    CTX(c);
    int q = 1, ch = c->B;
    if (use_fused(c) && !(c->slice && c->slice_ready)) {
        if (c->fs32 || c->fs64 || c->fused5) {
            const ChunkPlan p = chunk_plan(c->B, c->sched, c->fused5 != nullptr, c->fs64 != nullptr, c->sched.chunk_queues);
            q = p.queues; ch = p.chunk < c->B ? p.chunk : c->B;
        } else if (c->sched.chunk > 0) {
            ch = c->sched.chunk < c->B ? c->sched.chunk : c->B;
        } else if (c->sched.queues >= 2 && c->B >= 32 * c->sched.queues) {
            q = c->sched.queues;
        }
    }
    if (queues) *queues = q;
    if (chunk) *chunk = ch;
    if (launches_per_iteration) *launches_per_iteration = pnp_kernels_per_iteration(c);
    return PNP_OK;
}

int pnp_kernels_per_iteration(pnp_ctx* c) {
    if (!c) return 0;
    if (!use_fused(c)) return 3;                      // generic: rows, columns, rows
    if (c->slice && c->slice_ready) return 0;         // one launch per RUN: the iterations are a loop inside it
    if (c->fs32 || c->fs64 || c->fused5) {               // chunked round-robin schedules: two launches per chunk
        return chunk_plan_launches(c->B, chunk_plan(c->B, c->sched, c->fused5 != nullptr, c->fs64 != nullptr, c->sched.chunk_queues));
    }
    const int q = (c->sched.chunk > 0 || c->sched.queues < 2 || c->B < 32 * c->sched.queues) ? 1 : c->sched.queues;
    return 2 * q;                                     // two launches per queue and batched iteration
}
const char* pnp_path_name(pnp_ctx* c) {
    if (!c || !use_fused(c)) return "generic";
    return (c->slice && c->slice_ready) ? "slice" : "fused";
}

}  // extern "C"
