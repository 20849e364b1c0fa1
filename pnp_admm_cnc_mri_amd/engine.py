"""Host-side handle on one `pnp_ctx` of libpnpmri.so (include/pnp_mri.h).

`Engine` owns no numerics: every method is a thin call through the C ABI.  NumPy arrays are
passed as host pointers; objects exposing `data_ptr()` (torch CUDA tensors) as device pointers.
"""
import ctypes as C

import numpy as np

from . import _lib


def _is_dev(a):
    return hasattr(a, 'data_ptr')


def _ptr(a):
    if a is None:
        return None
    if _is_dev(a):
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(a.ctypes.data)


def _host(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


class Engine:
    """One context per (device, H, W, Bmax).  Not thread-safe, not re-entrant (as the ABI says)."""

    def __init__(self, H=256, W=256, Bmax=1, device=0, precision='f32'):
        """precision='f64': the reference's own arithmetic (S4:109) -- every buffer and step in double; problem
        (upload / synthesize), state, whole loops, x, metrics and SSIM are available, the step-wise
        float32 operators of the PnP path are not (pnp_mri.h, "double-precision context")."""
        if precision not in ('f32', 'f64'):
            raise ValueError("precision must be 'f32' or 'f64'")
        self._L = _lib.lib()
        self._ctx = _lib.ctx_p()
        self.f64 = precision == 'f64'
        create = self._L.pnp_ctx_create_f64 if self.f64 else self._L.pnp_ctx_create
        _lib.check(create(int(device), int(H), int(W), int(Bmax), C.byref(self._ctx)))
        self.H, self.W, self.Bmax, self.device = int(H), int(W), int(Bmax), int(device)
        self.B = 0
        self._real = np.float64 if self.f64 else np.float32
        self._cplx = np.complex128 if self.f64 else np.complex64

    # -- lifetime -------------------------------------------------------------------------
    def close(self):
        if getattr(self, '_ctx', None) is not None and self._ctx.value:
            self._L.pnp_ctx_destroy(self._ctx)
            self._ctx = _lib.ctx_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- plumbing -------------------------------------------------------------------------
    def set_stream(self, hip_stream):
        _lib.check(self._L.pnp_set_stream(self._ctx, C.c_void_p(int(hip_stream) if hip_stream else 0)))

    def sync(self):
        _lib.check(self._L.pnp_sync(self._ctx))

    def set_fast_path(self, enable):
        _lib.check(self._L.pnp_set_fast_path(self._ctx, 1 if enable else 0))

    def set_schedule(self, queues=2, mixed_launches=True, chunk=0):
        """Scheduling of the fused loops (bit-identical results for every setting); see pnp_set_schedule."""
        _lib.check(self._L.pnp_set_schedule(self._ctx, int(queues), 1 if mixed_launches else 0, int(chunk)))

    @property
    def schedule(self):
        q, m, ch = C.c_int(0), C.c_int(0), C.c_int(0)
        _lib.check(self._L.pnp_get_schedule(self._ctx, C.byref(q), C.byref(m), C.byref(ch)))
        return {'queues': q.value, 'mixed': m.value, 'chunk': ch.value}

    @property
    def plan(self):
        """What the next loop call does for the uploaded batch: {'queues', 'chunk', 'launches_per_iteration'} (pnp_get_plan)."""
        q, ch, n = C.c_int(), C.c_int(), C.c_int()
        _lib.check(self._L.pnp_get_plan(self._ctx, C.byref(q), C.byref(ch), C.byref(n)))
        return {'queues': q.value, 'chunk': ch.value, 'launches_per_iteration': n.value}

    @property
    def path_name(self):
        return self._L.pnp_path_name(self._ctx).decode()

    @property
    def kernels_per_iteration(self):
        return self._L.pnp_kernels_per_iteration(self._ctx)

    def timer_start(self):
        _lib.check(self._L.pnp_timer_start(self._ctx))

    def timer_stop(self):
        ms = C.c_float(0)
        _lib.check(self._L.pnp_timer_stop(self._ctx, C.byref(ms)))
        return ms.value

    # -- problem --------------------------------------------------------------------------
    def _masks(self, masks, mask_id, B):
        masks = np.asarray(masks)
        if masks.ndim == 2:
            masks = masks[None]
        if masks.shape[1:] != (self.H, self.W):
            raise ValueError('mask shape %s does not match engine %dx%d' % (masks.shape[1:], self.H, self.W))
        bank = _host(masks != 0, np.uint8)
        mid = None if mask_id is None else _host(mask_id, np.int32)
        if mid is not None and mid.shape != (B,):
            raise ValueError('mask_id must have shape (B,)')
        return bank, mid

    def upload(self, y, masks, mask_id=None):
        """y: [B,H,W] complex (host; cast to complex64) -- S4:102's `y`; masks [K,H,W] or [H,W]."""
        y = np.asarray(y)
        if y.ndim == 2:
            y = y[None]
        y = _host(y, self._cplx)
        B = y.shape[0]
        if y.shape[1:] != (self.H, self.W):
            raise ValueError('y shape %s does not match engine %dx%d' % (y.shape[1:], self.H, self.W))
        bank, mid = self._masks(masks, mask_id, B)
        up = self._L.pnp_upload_problem_f64 if self.f64 else self._L.pnp_upload_problem
        _lib.check(up(self._ctx, _ptr(y), _ptr(bank), _ptr(mid), B, bank.shape[0], 0))
        self.B = B

    def synthesize(self, img, noise, masks, mask_id=None):
        """y = fft2(img)*mask + noise on the device (S4:102).  img [B,H,W] float; noise [H,W]
        (shared, the reference's noises.mat) or [B,H,W] complex."""
        img = np.asarray(img)
        if img.ndim == 2:
            img = img[None]
        img = _host(img, np.float32)                      # the reference's img_L is float32 in either precision
        B = img.shape[0]
        noise = _host(noise, self._cplx)
        per = 1 if noise.ndim == 3 else 0
        if per and noise.shape[0] != B:
            raise ValueError('noise batch does not match images')
        if noise.shape[-2:] != (self.H, self.W):
            raise ValueError('noise shape %s does not match engine %dx%d' % (noise.shape[-2:], self.H, self.W))
        bank, mid = self._masks(masks, mask_id, B)
        fn = self._L.pnp_synthesize_problem_f64 if self.f64 else self._L.pnp_synthesize_problem
        _lib.check(fn(self._ctx, _ptr(img), _ptr(noise), per, _ptr(bank), _ptr(mid), B, bank.shape[0], 0))
        self.B = B

    def download_y(self):
        y = np.empty((self.B, self.H, self.W), self._cplx)
        fn = self._L.pnp_download_y_f64 if self.f64 else self._L.pnp_download_y
        _lib.check(fn(self._ctx, _ptr(y), 0))
        return y

    def init_state(self):
        _lib.check(self._L.pnp_init_state(self._ctx))

    def prepare_loops(self):
        """Build now the per-problem tables the whole loops will use (otherwise built by the first loop call)."""
        _lib.check(self._L.pnp_prepare_loops(self._ctx))

    def set_state(self, z=None, w=None):
        zz = None if z is None else (z if _is_dev(z) else _host(z, self._real))
        ww = None if w is None else (w if _is_dev(w) else _host(w, self._real))
        dev = 1 if (_is_dev(z) or _is_dev(w)) else 0
        fn = self._L.pnp_set_state_f64 if self.f64 else self._L.pnp_set_state
        _lib.check(fn(self._ctx, _ptr(zz), _ptr(ww), dev))

    def get_state(self, z_out=None, w_out=None):
        """-> (z, w) as host arrays, or copied device-to-device into the given device tensors."""
        fn = self._L.pnp_get_state_f64 if self.f64 else self._L.pnp_get_state
        if z_out is not None or w_out is not None:
            if not all(o is None or _is_dev(o) for o in (z_out, w_out)):
                raise TypeError('get_state outputs must be device tensors')
            _lib.check(fn(self._ctx, _ptr(z_out), _ptr(w_out), 1))
            return z_out, w_out
        z = np.empty((self.B, self.H, self.W), self._real)
        w = np.empty_like(z)
        _lib.check(fn(self._ctx, _ptr(z), _ptr(w), 0))
        return z, w

    # -- whole loops ----------------------------------------------------------------------
    def admm_l1(self, iters, lambda1, reo):
        _lib.check(self._L.pnp_admm_l1_run(self._ctx, int(iters), float(lambda1), float(reo)))

    def admm_cnc(self, iters, alpha, lambda1, reo, b):
        _lib.check(self._L.pnp_admm_cnc_run(self._ctx, int(iters), float(alpha), float(lambda1), float(reo), float(b)))

    def x(self, out=None):
        fn = self._L.pnp_download_x_f64 if self.f64 else self._L.pnp_download_x
        if out is not None and _is_dev(out):
            _lib.check(fn(self._ctx, _ptr(out), 1))
            return out
        x = np.empty((self.B, self.H, self.W), self._real)
        _lib.check(fn(self._ctx, _ptr(x), 0))
        return x

    # -- step-wise operators on device tensors (PnP path) ---------------------------------
    def dc_step(self, z, w, x, reo):
        _lib.check(self._L.pnp_dc_step(self._ctx, _ptr(z), _ptr(w), _ptr(x), float(reo)))

    def prox_l1_dual(self, x, z, w, thr):
        _lib.check(self._L.pnp_prox_l1_dual(self._ctx, _ptr(x), _ptr(z), _ptr(w), float(thr)))

    def prox_cnc_dual(self, x, z, w, alpha, lambda1, reo, b):
        _lib.check(self._L.pnp_prox_cnc_dual(self._ctx, _ptr(x), _ptr(z), _ptr(w), float(alpha), float(lambda1),
                                             float(reo), float(b)))

    def cnc_combine(self, z, x, w, s, t, alpha, lambda1, reo, b):
        _lib.check(self._L.pnp_cnc_combine(self._ctx, _ptr(z), _ptr(x), _ptr(w), _ptr(s), _ptr(t), float(alpha),
                                           float(lambda1), float(reo), float(b)))

    def add(self, a, b, out):
        _lib.check(self._L.pnp_add(self._ctx, _ptr(a), _ptr(b), _ptr(out)))

    def dual_clamp(self, x, z, w):
        _lib.check(self._L.pnp_dual_clamp(self._ctx, _ptr(x), _ptr(z), _ptr(w)))

    # -- operator API on device tensors ---------------------------------------------------
    def fft2(self, inp, out, B):
        _lib.check(self._L.pnp_fft2_fwd(self._ctx, _ptr(inp), _ptr(out), int(B)))

    def ifft2(self, inp, out, B):
        _lib.check(self._L.pnp_fft2_inv(self._ctx, _ptr(inp), _ptr(out), int(B)))

    def A(self, x, k):
        _lib.check(self._L.pnp_A(self._ctx, _ptr(x), _ptr(k)))

    def AH(self, k, out):
        _lib.check(self._L.pnp_AH(self._ctx, _ptr(k), _ptr(out)))

    def Df(self, x, out):
        _lib.check(self._L.pnp_Df(self._ctx, _ptr(x), _ptr(out)))

    def metrics(self, x_dev, gt_u8):
        """-> (psnr[B], re[B]) of img_E = x*255 against uint8 ground truth
        (utils/utils_image.py:543-556, 622-636)."""
        gt = _host(gt_u8, np.uint8)
        psnr = np.empty(self.B, np.float64)
        re = np.empty(self.B, np.float64)
        fn = self._L.pnp_metrics_f64 if self.f64 else self._L.pnp_metrics
        _lib.check(fn(self._ctx, _ptr(x_dev), _ptr(gt), 0,
                      psnr.ctypes.data_as(_lib.c_double_p), re.ctypes.data_as(_lib.c_double_p)))
        return psnr, re

    def ssim(self, x_dev, gt_u8):
        """-> ssim[B] of img_E = x*255 against uint8 ground truth (utils/utils_image.py:570-615), on device."""
        gt = _host(gt_u8, np.uint8)
        out = np.empty(self.B, np.float64)
        fn = self._L.pnp_ssim_f64 if self.f64 else self._L.pnp_ssim
        _lib.check(fn(self._ctx, _ptr(x_dev), _ptr(gt), 0, out.ctypes.data_as(_lib.c_double_p)))
        return out
