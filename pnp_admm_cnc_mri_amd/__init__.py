"""pnp_admm_cnc_mri_amd -- MI355X-native PnP-ADMM-CNC MRI reconstruction.

Host code in Python (mirroring the reference's solver functions and utils_pnp API), compute in
hand-written HIP kernels for gfx950 behind the C ABI of include/pnp_mri.h (libpnpmri.so).
"""
from .engine import Engine                                    # noqa: F401
from .solvers import ADMM_L1, ADMM_CNC, PRESETS               # noqa: F401
from . import utils_pnp                                       # noqa: F401

__all__ = ['Engine', 'ADMM_L1', 'ADMM_CNC', 'PRESETS', 'utils_pnp']
