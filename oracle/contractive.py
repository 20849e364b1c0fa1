"""TEST INFRASTRUCTURE: measurements behind the contractive fixture weights (pnp_admm_cnc_mri_amd.denoisers.contractive_state_dict).

  conv_operator_norms(module, seed)   the operator norm of every convolution of `module` with the seeded standard-normal kernel R of
                                      its state_dict key (power iteration on conv^T conv, float64, CPU) -- the `gains` committed in
                                      tests/golden/pnp_known.json
  lipschitz_at(fn, x, iters)          largest singular value of the Jacobian of `fn` at x (power iteration through autograd): the
                                      local Lipschitz constant recorded beside the goldens

Used by oracle/make_golden_pnp.py (the container, CPU); nothing of the product imports it.
"""
import hashlib

import numpy as np
import torch
import torch.nn.functional as F


def _seeded_normal(seed, key, shape):
    h = int.from_bytes(hashlib.sha256(('%d:%s' % (seed, key)).encode()).digest()[:8], 'little')
    return np.random.default_rng(h).standard_normal(tuple(shape))


def _op_norm(apply, apply_t, in_shape, iters=60, seed=0):
    g = torch.Generator().manual_seed(seed)
    v = torch.randn(in_shape, dtype=torch.float64, generator=g)
    v /= v.norm()
    s = 0.0
    for _ in range(iters):
        u = apply(v)
        s = float(u.norm())
        v = apply_t(u)
        v /= v.norm()
    return s


def conv_operator_norms(module, seed, size=24):
    """{weight key: ||conv(R_key)||} for every Conv2d / ConvTranspose2d of `module` (stride, padding and dilation as declared, zero
    padding, `size` x `size` inputs: the norm of a zero-padded convolution grows with the size towards the circular one's and is
    within a per cent of it at 24 pixels for 3 x 3 kernels)."""
    mods = dict(module.named_modules())
    out = {}
    for key, v in module.state_dict().items():
        if not key.endswith('weight'):
            continue
        m = mods[key[:-len('.weight')]]
        r = torch.from_numpy(_seeded_normal(seed, key, v.shape))
        if isinstance(m, torch.nn.ConvTranspose2d):
            kw = dict(stride=m.stride, padding=m.padding)
            n_in = m.in_channels
            sp = max(4, size // 2)
            ap = lambda x, r=r, kw=kw: F.conv_transpose2d(x, r, **kw)
            at = lambda y, r=r, kw=kw: F.conv2d(y, r, **kw)
        else:
            kw = dict(stride=m.stride, padding=m.padding, dilation=m.dilation)
            n_in = m.in_channels
            sp = size + 2 * (m.dilation[0] - 1) * 2
            shape_in = (1, n_in, sp, sp)
            ap = lambda x, r=r, kw=kw: F.conv2d(x, r, **kw)

            def at(y, r=r, kw=kw, shape_in=shape_in):
                return torch.nn.grad.conv2d_input(shape_in, r, y, **kw)
        out[key] = _op_norm(ap, at, (1, n_in, sp, sp))
    return out


def lipschitz_at(fn, x, iters=25, seed=1):
    """sigma_max of d fn / d x at x (float32 CPU tensors; fn differentiable almost everywhere)."""
    g = torch.Generator().manual_seed(seed)
    v = torch.randn(x.shape, generator=g)
    v /= v.norm()
    s = 0.0
    for _ in range(iters):
        xx = x.clone().requires_grad_(True)
        with torch.enable_grad():
            _, jv = torch.autograd.functional.jvp(fn, xx, v)
            s = float(jv.norm())
            _, vj = torch.autograd.functional.vjp(fn, xx, jv)
        v = vj / vj.norm()
    return s
