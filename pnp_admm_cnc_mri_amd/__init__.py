"""pnp_admm_cnc_mri_amd -- MI355X-native PnP-ADMM-CNC MRI reconstruction.

Host code in Python (mirroring the reference's solver functions and utils_pnp API), compute in
hand-written HIP kernels for gfx950 behind the C ABI of include/pnp_mri.h (libpnpmri.so);
PyTorch-ROCm only for the denoiser forward pass, device tensors and torch.distributed.
"""
from .engine import Engine                                    # noqa: F401
from .solvers import ADMM_L1, ADMM_CNC, PRESETS               # noqa: F401
from . import utils_pnp                                       # noqa: F401


def __getattr__(name):
    # the PnP entry points pull in torch.nn; import them on first use
    if name in ('PNP_ADMM_L1_D', 'PNP_ADMM_CNC_D', 'PNP_ADMM_CNC_DnCNN', 'solvers_pnp', 'denoisers', 'sharding'):
        import importlib
        if name in ('solvers_pnp', 'denoisers', 'sharding'):
            return importlib.import_module('.' + name, __name__)
        return getattr(importlib.import_module('.solvers_pnp', __name__), name)
    raise AttributeError(name)


__all__ = ['Engine', 'ADMM_L1', 'ADMM_CNC', 'PNP_ADMM_L1_D', 'PNP_ADMM_CNC_D', 'PNP_ADMM_CNC_DnCNN', 'PRESETS',
           'utils_pnp']
