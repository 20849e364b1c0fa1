"""ctypes binding of libpnpmri.so (C ABI: include/pnp_mri.h).

The product path has no CPU fallback: if the HIP library is missing or fails to load, `lib()`
raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C pnp_admm_cnc_mri_amd/csrc`.
"""
import ctypes as C
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# PNP_MRI_LIB: developer knob for A/B runs of two builds of the library; the product ships the in-tree one
LIB_PATH = os.environ.get('PNP_MRI_LIB') or os.path.join(_HERE, 'libpnpmri.so')

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
ctx_p = C.c_void_p
_vp = C.c_void_p

# name -> (restype, argtypes); mirrors include/pnp_mri.h one to one
SIGNATURES = {
    'pnp_abi_version': (C.c_int, []),
    'pnp_device_info': (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int, C.c_char_p, C.c_int]),
    'pnp_calibrate_stream': (C.c_int, [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_double)]),
    'pnp_last_error': (C.c_char_p, []),
    'pnp_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'pnp_ctx_create': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(ctx_p)]),
    'pnp_ctx_destroy': (C.c_int, [ctx_p]),
    'pnp_set_stream': (C.c_int, [ctx_p, _vp]),
    'pnp_sync': (C.c_int, [ctx_p]),
    'pnp_set_fast_path': (C.c_int, [ctx_p, C.c_int]),
    'pnp_set_schedule': (C.c_int, [ctx_p, C.c_int, C.c_int, C.c_int]),
    'pnp_get_schedule': (C.c_int, [ctx_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'pnp_get_plan': (C.c_int, [ctx_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'pnp_upload_problem': (C.c_int, [ctx_p, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int]),
    'pnp_synthesize_problem': (C.c_int, [ctx_p, _vp, _vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int]),
    'pnp_download_y': (C.c_int, [ctx_p, _vp, C.c_int]),
    'pnp_init_state': (C.c_int, [ctx_p]),
    'pnp_set_state': (C.c_int, [ctx_p, _vp, _vp, C.c_int]),
    'pnp_get_state': (C.c_int, [ctx_p, _vp, _vp, C.c_int]),
    'pnp_admm_l1_run': (C.c_int, [ctx_p, C.c_int, C.c_double, C.c_double]),
    'pnp_admm_cnc_run': (C.c_int, [ctx_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]),
    'pnp_download_x': (C.c_int, [ctx_p, _vp, C.c_int]),
    'pnp_dc_step': (C.c_int, [ctx_p, _vp, _vp, _vp, C.c_double]),
    'pnp_prox_l1_dual': (C.c_int, [ctx_p, _vp, _vp, _vp, C.c_double]),
    'pnp_prox_cnc_dual': (C.c_int, [ctx_p, _vp, _vp, _vp, C.c_double, C.c_double, C.c_double, C.c_double]),
    'pnp_cnc_combine': (C.c_int, [ctx_p, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_double, C.c_double]),
    'pnp_add': (C.c_int, [ctx_p, _vp, _vp, _vp]),
    'pnp_dual_clamp': (C.c_int, [ctx_p, _vp, _vp, _vp]),
    'pnp_fft2_fwd': (C.c_int, [ctx_p, _vp, _vp, C.c_int]),
    'pnp_fft2_inv': (C.c_int, [ctx_p, _vp, _vp, C.c_int]),
    'pnp_A': (C.c_int, [ctx_p, _vp, _vp]),
    'pnp_AH': (C.c_int, [ctx_p, _vp, _vp]),
    'pnp_Df': (C.c_int, [ctx_p, _vp, _vp]),
    'pnp_metrics': (C.c_int, [ctx_p, _vp, _vp, C.c_int, c_double_p, c_double_p]),
    'pnp_ctx_create_f64': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(ctx_p)]),
    'pnp_upload_problem_f64': (C.c_int, [ctx_p, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int]),
    'pnp_set_state_f64': (C.c_int, [ctx_p, _vp, _vp, C.c_int]),
    'pnp_get_state_f64': (C.c_int, [ctx_p, _vp, _vp, C.c_int]),
    'pnp_download_x_f64': (C.c_int, [ctx_p, _vp, C.c_int]),
    'pnp_is_f64': (C.c_int, [ctx_p]),
    'pnp_synthesize_problem_f64': (C.c_int, [ctx_p, _vp, _vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int]),
    'pnp_download_y_f64': (C.c_int, [ctx_p, _vp, C.c_int]),
    'pnp_metrics_f64': (C.c_int, [ctx_p, _vp, _vp, C.c_int, c_double_p, c_double_p]),
    'pnp_ssim_f64': (C.c_int, [ctx_p, _vp, _vp, C.c_int, c_double_p]),
    'pnp_prepare_loops': (C.c_int, [ctx_p]),
    'pnp_conv3x3_c64_nhwc': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'pnp_conv3x3_c64_pack': (C.c_int, [_vp, _vp, _vp]),
    'pnp_conv3x3_c64_nhwc_f16x3': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'pnp_conv3x3_c64_pack_f16x3': (C.c_int, [_vp, _vp, _vp]),
    'pnp_conv3x3_nhwc_f16x3': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'pnp_conv3x3_f16x3_set_variant': (C.c_int, [C.c_int]),
    'pnp_conv3x3_nhwc_f16x3_fmt': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'pnp_conv3x3_pack_f16x3': (C.c_int, [_vp, _vp, _vp, C.c_int]),
    'pnp_conv3x3_tail_nchw_f16x3': (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    'pnp_conv3x3_tail_add_nchw_f16x3': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    'pnp_conv2x2s2_nhwc_f16x3': (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    'pnp_convT2x2s2_nhwc_f16x3': (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    'pnp_conv2x2_pack_f16x3': (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int]),
    'pnp_ffdnet_head_nhwc': (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    'pnp_ffdnet_tail_f16x3': (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int]),
    'pnp_conv3x3_head_nhwc': (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'pnp_conv3x3_tail_nchw': (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    'pnp_relayout_c64': (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    'pnp_ssim': (C.c_int, [ctx_p, _vp, _vp, C.c_int, c_double_p]),
    'pnp_timer_start': (C.c_int, [ctx_p]),
    'pnp_timer_stop': (C.c_int, [ctx_p, c_float_p]),
    'pnp_kernels_per_iteration': (C.c_int, [ctx_p]),
    'pnp_path_name': (C.c_char_p, [ctx_p]),
}

ABI_VERSION = 11
_lib = None


class PnpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('libpnpmri error %d: %s' % (code, msg))
        self.code = code


def lib():
    """Load libpnpmri.so once; raise (never fall back) if it is absent or ABI-mismatched."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError('%s not found: build the HIP extension first (python -c "import __graft_entry__ as g; '
                          'g.build()" or make -C pnp_admm_cnc_mri_amd/csrc); there is no CPU fallback' % LIB_PATH)
    # One HIP runtime per process: libpnpmri.so and PyTorch-ROCm both need "libamdhip64.so.7" and
    # the dynamic loader keeps whichever copy arrives first.  The PnP solvers share streams and
    # device pointers with torch, so torch's bundled runtime must be that copy -- load it first.
    if 'torch' not in sys.modules and importlib.util.find_spec('torch') is not None:
        import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)           # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if L.pnp_abi_version() != ABI_VERSION:
        raise ImportError('libpnpmri.so ABI %d, binding expects %d' % (L.pnp_abi_version(), ABI_VERSION))
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise PnpError(rc, lib().pnp_last_error().decode('utf-8', 'replace'))


def device_count():
    n = C.c_int(0)
    check(lib().pnp_device_count(C.byref(n)))
    return n.value
