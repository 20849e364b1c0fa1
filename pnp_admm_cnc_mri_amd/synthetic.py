"""Deterministic synthetic workload of SURVEY.md section 8(d): seeded ellipse phantoms, k-space
noise (std 15 per component, on all points) and seeded 30 % masks for sizes the reference ships
no .mat for.  Used by bench.py and smoke(); tests/test_synthetic.py checks it against the
oracle's independent copy so CPU baseline and GPU see identical inputs."""
import numpy as np


def phantom(b, H=256, W=256):
    rng = np.random.default_rng(20260000 + b)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    img = np.zeros((H, W))
    for _ in range(12):
        cy, cx = rng.uniform(0.2, 0.8, 2) * (H, W)
        ry, rx = rng.uniform(0.05, 0.35, 2) * (H, W)
        amp = rng.uniform(0.1, 0.5)
        th = rng.uniform(0, np.pi)
        u = ((yy - cy) * np.cos(th) + (xx - cx) * np.sin(th)) / ry
        v = (-(yy - cy) * np.sin(th) + (xx - cx) * np.cos(th)) / rx
        r = np.sqrt(u * u + v * v)
        img += amp * np.clip((1.0 - r) * 8.0, 0.0, 1.0)
    img = np.clip(img, 0, 1)
    return np.float32(np.round(img * 255.) / 255.)


def kspace_noise(b, H=256, W=256, std=15.0):
    rng = np.random.default_rng(777 + b)
    return std * (rng.standard_normal((H, W)) + 1j * rng.standard_normal((H, W)))


def synthetic_mask(kind, H, W, rate=0.30, seed=4242):
    rng = np.random.default_rng(seed + {'random': 0, 'radial': 1, 'cartesian': 2}[kind])
    fy = np.fft.fftfreq(H)[:, None]
    fx = np.fft.fftfreq(W)[None, :]
    if kind == 'random':
        r = np.sqrt(fy ** 2 + fx ** 2) / 0.5
        pdf = (1 - np.clip(r, 0, 1)) ** 3 + 0.02
        lo, hi = 0.0, 50.0
        for _ in range(60):
            s = 0.5 * (lo + hi)
            if np.minimum(pdf * s, 1).mean() > rate:
                hi = s
            else:
                lo = s
        m = rng.uniform(size=(H, W)) < np.minimum(pdf * s, 1)
    elif kind == 'radial':
        m = np.zeros((H, W), bool)
        nlines = int(rate * np.pi * min(H, W) / 2 * 0.62)
        t = np.linspace(-0.5, 0.5, 4 * max(H, W))
        for a in np.arange(nlines) * np.pi / nlines:
            iy = np.round(t * np.sin(a) * H).astype(int) % H
            ix = np.round(t * np.cos(a) * W).astype(int) % W
            m[iy, ix] = True
    else:
        m = np.zeros((H, W), bool)
        centre = int(0.08 * H)
        rows = np.abs(np.fft.fftfreq(H) * H) <= centre / 2
        rest = np.flatnonzero(~rows)
        take = rng.choice(rest, size=max(int(rate * H) - rows.sum(), 0), replace=False)
        rows[take] = True
        m[rows, :] = True
    m[0, 0] = True
    return m.astype(np.uint8)


def batch(first, B, H=256, W=256):
    """-> (img float32 [B,H,W], noise complex64 [B,H,W]) for slices first .. first+B-1."""
    img = np.empty((B, H, W), np.float32)
    noise = np.empty((B, H, W), np.complex64)
    for i in range(B):
        img[i] = phantom(first + i, H, W)
        noise[i] = kspace_noise(first + i, H, W)
    return img, noise


_MASK_FIXTURE = None


def reference_masks():
    """The reference's three 256x256 sampling patterns (CS_MRI/Q_{Random,Radial,Cartesian}30.mat,
    variable Q1) as package data, bit-packed (pnp_admm_cnc_mri_amd/data/cs_mri_masks.npz, 8 KiB each)."""
    global _MASK_FIXTURE
    if _MASK_FIXTURE is None:
        import os
        d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'cs_mri_masks.npz'))
        _MASK_FIXTURE = {k[:-len('_packbits')]: np.unpackbits(d[k])[:65536].reshape(256, 256)
                         for k in d.files if k.endswith('_packbits')}
    return _MASK_FIXTURE
