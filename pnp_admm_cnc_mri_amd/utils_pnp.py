"""`utils_pnp` of the reference (utils/utils_pnp.py) -- same names, same arguments -- plus the
operator API the north star asks for next to it (A, AH, Df, dc_solve, prox_l1, prox_cnc), all
batched and running on the HIP engine.

The two schedule functions are host scalar code (iter_num floats); they follow
utils/utils_pnp.py:14-34 and are pinned bit-for-bit by tests/test_oracle_golden.py.
"""
import numpy as np


def get_rho_sigma(sigma=2.55 / 255, iter_num=15, modelSigma1=49.0, modelSigma2=2.55, w=1.0):
    """utils/utils_pnp.py:14-23: log/linear blend of the denoiser noise levels, and rho_k."""
    modelSigmaS = np.logspace(np.log10(modelSigma1), np.log10(modelSigma2), iter_num).astype(np.float32)
    modelSigmaS_lin = np.linspace(modelSigma1, modelSigma2, iter_num).astype(np.float32)
    sigmas = (modelSigmaS * w + modelSigmaS_lin * (1 - w)) / 255.
    rhos = list(map(lambda x: 0.23 * (sigma ** 2) / (x ** 2), sigmas))
    return rhos, sigmas


def get_rho_sigma1(sigma=2.55 / 255, iter_num=15, modelSigma1=49.0, modelSigma2=2.55, lamda=3.0):
    """utils/utils_pnp.py:26-34."""
    modelSigmaS = np.logspace(np.log10(modelSigma1), np.log10(modelSigma2), iter_num).astype(np.float32)
    sigmas = modelSigmaS / 255.
    rhos = list(map(lambda x: (sigma ** 2) / (x ** 2) / lamda, sigmas))
    return rhos, sigmas


# ----------------------------------------------------------------------------------------------
# operator API: torch CUDA tensors in, torch CUDA tensors out; an `Engine` holds y and the masks
# ----------------------------------------------------------------------------------------------
def _torch():
    import torch
    return torch


def _prep(eng, t):
    torch = _torch()
    if not t.is_cuda:
        raise ValueError('operator API works on CUDA tensors (no CPU fallback)')
    eng.set_stream(torch.cuda.current_stream(t.device).cuda_stream)
    return t.contiguous()


def fft2(eng, x):
    """np.fft.fft2 of a complex64 batch [B,H,W] (S4:102, 120)."""
    torch = _torch()
    x = _prep(eng, x.to(torch.complex64))
    out = torch.empty_like(x)
    eng.fft2(torch.view_as_real(x), torch.view_as_real(out), x.shape[0])
    return out


def ifft2(eng, k):
    """np.fft.ifft2 (S4:103, 123)."""
    torch = _torch()
    k = _prep(eng, k.to(torch.complex64))
    out = torch.empty_like(k)
    eng.ifft2(torch.view_as_real(k), torch.view_as_real(out), k.shape[0])
    return out


def A(eng, x):
    """A x = fft2(x) * mask for real x [B,H,W] with the engine's uploaded masks (S4:102)."""
    torch = _torch()
    x = _prep(eng, x.float())
    k = torch.empty(x.shape, dtype=torch.complex64, device=x.device)
    eng.A(x, torch.view_as_real(k))
    return k


def AH(eng, k):
    """A^H k = ifft2(k * mask) (utils/utils.py:54)."""
    torch = _torch()
    k = _prep(eng, k.to(torch.complex64))
    out = torch.empty_like(k)
    eng.AH(torch.view_as_real(k), torch.view_as_real(out))
    return out


def Df(eng, x):
    """Df(x, mask, y) = A^H (A x - y) of utils/utils.py:50-55, with the engine's y and masks."""
    torch = _torch()
    x = _prep(eng, x.float())
    out = torch.empty(x.shape, dtype=torch.complex64, device=x.device)
    eng.Df(x, torch.view_as_real(out))
    return out


def dc_solve(eng, z, w, reo):
    """x-update x = |Re ifft2((La2*F(z-w) + M^T y)/(La2 + M))|, La2 = 1/(2 reo) (S4:119-124)."""
    torch = _torch()
    z = _prep(eng, z.float())
    w = _prep(eng, w.float())
    x = torch.empty_like(z)
    eng.dc_step(z, w, x, reo)
    return x


def prox_l1(eng, x, z, w, thr):
    """in place: z = soft(x + w, thr); w = w + x - z (S1:123, 126).  Returns (z, w)."""
    _prep(eng, x)
    eng.prox_l1_dual(x, z, w, thr)
    return z, w


def prox_cnc(eng, x, z, w, alpha, lambda1, reo, b):
    """in place CNC z-update + dual update (S4:127-132).  Returns (z, w)."""
    _prep(eng, x)
    eng.prox_cnc_dual(x, z, w, alpha, lambda1, reo, b)
    return z, w
