"""Host-side image metrics with the reference's definitions (utils/utils_image.py:543-636).
PSNR and RE for whole batches are computed on the device (pnp_metrics in include/pnp_mri.h);
the functions here serve single images and SSIM (off the hot loop, per image)."""
import math

import numpy as np


def calculate_psnr(img1, img2, border=0):
    """img1, img2 in [0,255]; utils_image.py:543-556."""
    if not img1.shape == img2.shape:
        raise ValueError('Input images must have the same dimensions.')
    h, w = img1.shape[:2]
    a = img1[border:h - border, border:w - border].astype(np.float64)
    b = img2[border:h - border, border:w - border].astype(np.float64)
    mse = np.mean((a - b) ** 2)
    if mse == 0:
        return float('inf')
    return 20 * math.log10(255.0 / math.sqrt(mse))


def calculate_re(img1, img2, border=0):
    """||img2 - img1|| / ||img2||; utils_image.py:622-636."""
    if not img1.shape == img2.shape:
        raise ValueError('Input images must have the same dimensions.')
    h, w = img1.shape[:2]
    a = img1[border:h - border, border:w - border].astype(np.float64)
    b = img2[border:h - border, border:w - border].astype(np.float64)
    return float(np.linalg.norm(b - a) / np.linalg.norm(b))


def psnr(x, im_orig):
    """max = 255 variant that accepts complex input ("zero-filling psnr", S4:104)."""
    M, N = np.shape(x)
    mse = (np.sum((np.absolute(x - im_orig)) ** 2)) / (M * N)
    return 10 * np.log10(255 * 255 / mse)


def _sep_valid(a, k):
    """separable 'valid' correlation with the symmetric 1-D kernel k (11 taps)."""
    n = len(k)
    tmp = np.zeros((a.shape[0], a.shape[1] - n + 1))
    for i in range(n):
        tmp += k[i] * a[:, i:i + tmp.shape[1]]
    out = np.zeros((a.shape[0] - n + 1, tmp.shape[1]))
    for i in range(n):
        out += k[i] * tmp[i:i + out.shape[0], :]
    return out


def calculate_ssim(img1, img2, border=0):
    """Gaussian 11/1.5 window on the valid region (the reference crops filter2D's output by
    [5:-5], so its border mode never matters); utils_image.py:570-615, gray images."""
    if not img1.shape == img2.shape:
        raise ValueError('Input images must have the same dimensions.')
    h, w = img1.shape[:2]
    a = np.squeeze(img1[border:h - border, border:w - border]).astype(np.float64)
    b = np.squeeze(img2[border:h - border, border:w - border]).astype(np.float64)
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    g = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))
    g /= g.sum()
    mu1, mu2 = _sep_valid(a, g), _sep_valid(b, g)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 ** 2, mu2 ** 2, mu1 * mu2
    s1 = _sep_valid(a * a, g) - mu1_sq
    s2 = _sep_valid(b * b, g) - mu2_sq
    s12 = _sep_valid(a * b, g) - mu1_mu2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return float(ssim_map.mean())
