"""CPU oracle for the PnP-ADMM hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``pnp_admm_cnc_mri_amd``) never does: it calls the HIP
kernels through the C-ABI of ``include/pnp_mri.h`` and fails loudly when that library is missing.

It is a plain NumPy restatement (float64, ``np.fft`` = pocketfft) of the per-iteration loop that
is copy-pasted inline in every solver script of zj15001/PNP_ADMM_CNC_MRI.  Citations use the
aliases of SURVEY.md: S1 = ``【1】ADMM_L1.py``, S3 = ``【3】PNP_ADMM_L1_D  .py``,
S4 = ``【4】ADMM_CNC .py``, S6 = ``【6】PNP_ADMM_CNC_D .py``.

Parity pin: ``tests/test_oracle_golden.py`` checks every function below against vectors produced
by the *unmodified* reference scripts in the build container (``oracle/make_golden.py``; NumPy
2.2.6) and against the scalar known answers in the reference's own logs
(``results/Set1_dn_ADMM_L1/Set1_dn_ADMM_L1.log:284-288``,
``results/Set1_dn_ADMM_CNC/Set1_dn_ADMM_CNC.log:399-400``).

dtype semantics follow NumPy >= 2 exactly as the reference experiences them: the image is
float32 (utils/utils_image.py:181-182) so the first ``fft2`` runs in complex64 (S4:102) and is
promoted to complex128 by the float64 mask / complex128 noise; everything after is float64.
"""
import numpy as np


# ----------------------------------------------------------------------------------------------
# a1  soft()                                                              S1:18-19 == S4:18-19
# ----------------------------------------------------------------------------------------------
def soft(x, c):
    return np.fmax(np.fabs(x) - c, 0) * np.sign(x)


# ----------------------------------------------------------------------------------------------
# a2  measurement synthesis + zero-filled initialisation     S4:101-109 == S1:97-105, S6:250-256
# ----------------------------------------------------------------------------------------------
def requantise(img_uint8):
    """uint8 gray image -> the float32 image the solvers see (S4:91-94:
    ``uint2single(single2uint(uint2single(img)))``; utils/utils_image.py:181-186)."""
    img = np.float32(np.asarray(img_uint8) / 255.)
    img = np.uint8((img.clip(0, 1) * 255.).round())
    return np.float32(img / 255.)


def synthesize(img_L, mask, noises):
    """y = fft2(img_L) * mask + noises  (S4:102).  Noise is added on ALL k-space points."""
    return np.fft.fft2(img_L) * mask + noises


def init_state(y, real=np.float64):
    """x0 = |ifft2(y)| (complex modulus), z0 = x0, w0 = 0   (S4:103-109).
    ``real=np.float32`` is the precision control used by tests (see admm_*_f32 below)."""
    x = np.absolute(np.fft.ifft2(y))
    z = np.copy(x)
    w = np.zeros(y.shape, dtype=real)
    return x, z, w


# ----------------------------------------------------------------------------------------------
# a3  x-update / data-consistency solve in k-space       S4:119-124 == S1:115-120 == S6:266-271
# ----------------------------------------------------------------------------------------------
def dc_step(z, w, y, mask, reo):
    index = np.nonzero(mask)
    xtilde = np.copy(z - w)
    xf = np.fft.fft2(xtilde)
    La2 = 1.0 / 2.0 / reo
    xf[index] = (La2 * xf[index] + y[index]) / (1.0 + La2)
    x = np.real(np.fft.ifft2(xf))
    return np.absolute(x)


# ----------------------------------------------------------------------------------------------
# a4/a6  L1 z-update + dual update                                                 S1:123, S1:126
# ----------------------------------------------------------------------------------------------
def l1_step(x, z, w, lambda1, reo):
    z = soft(x + w, reo * lambda1)
    w = w + x - z
    return z, w


# ----------------------------------------------------------------------------------------------
# a5/a6  CNC z-update + dual update                                            S4:127-129, S4:132
# ----------------------------------------------------------------------------------------------
def cnc_step(x, z, w, alpha, lambda1, reo, b):
    s = soft(z, 1 / b)
    t = (1 - alpha) * z + alpha * (x + w) + alpha * reo * lambda1 * b * (z - s)
    z = soft(t, alpha * reo * lambda1)
    w = w + x - z
    return z, w


# ----------------------------------------------------------------------------------------------
# whole loops; ``trace`` = iteration numbers (1-based) at which (x, z, w) are recorded
# ----------------------------------------------------------------------------------------------
def admm_l1(y, mask, iter_num=50, lambda1=0.1, reo=0.015, trace=()):
    """S1:105-126 for one slice.  Defaults are the committed CLI presets (S1:171)."""
    x, z, w = init_state(y)
    rec = {}
    for i in range(iter_num):
        x = dc_step(z, w, y, mask, reo)
        z, w = l1_step(x, z, w, lambda1, reo)
        if (i + 1) in trace:
            rec[i + 1] = (x.copy(), z.copy(), w.copy())
    return (x, rec) if trace else x


def admm_cnc(y, mask, iter_num=50, alpha=0.45, lambda1=0.5, reo=0.05, b=64, trace=()):
    """S4:107-132 for one slice.  Defaults are the committed CLI presets (S4:176)."""
    x, z, w = init_state(y)
    rec = {}
    for i in range(iter_num):
        x = dc_step(z, w, y, mask, reo)
        z, w = cnc_step(x, z, w, alpha, lambda1, reo, b)
        if (i + 1) in trace:
            rec[i + 1] = (x.copy(), z.copy(), w.copy())
    return (x, rec) if trace else x


# ----------------------------------------------------------------------------------------------
# Precision controls (NOT reference behaviour): the same NumPy lines run with complex64 / float32
# arrays.  The committed CNC presets make the iteration map locally expansive (SURVEY.md section
# 7), so fp32 round-off grows by ~1.08x per iteration in ANY fp32 implementation; tests use these
# to state "the HIP path deviates from the float64 reference no more than NumPy's own float32
# arithmetic does" where the north star's 1e-5 is unreachable in fp32 (e.g. 100 CNC iterations).
# ----------------------------------------------------------------------------------------------
def admm_l1_f32(y, mask, iter_num=50, lambda1=0.1, reo=0.015):
    y = y.astype(np.complex64)
    x, z, w = init_state(y, np.float32)
    for i in range(iter_num):
        x = dc_step(z, w, y, mask, reo)
        z, w = l1_step(x, z, w, lambda1, reo)
    assert x.dtype == np.float32 and z.dtype == np.float32 and w.dtype == np.float32
    return x


def admm_cnc_f32(y, mask, iter_num=50, alpha=0.45, lambda1=0.5, reo=0.05, b=64):
    y = y.astype(np.complex64)
    x, z, w = init_state(y, np.float32)
    for i in range(iter_num):
        x = dc_step(z, w, y, mask, reo)
        z, w = cnc_step(x, z, w, alpha, lambda1, reo, b)
    assert x.dtype == np.float32 and z.dtype == np.float32 and w.dtype == np.float32
    return x


# ----------------------------------------------------------------------------------------------
# a8-a10  PnP variants.  ``denoise(t, i)`` maps a float32 [H,W] array to a float32 [H,W] array
# (the CNN; the reference runs it on [1,1,H,W] tensors).  The marshalling side effects of
# S6:273-285 / S6:305-308 are semantics: |z|, |w| before the step, float32 state, clamp(0,1) of
# x, z AND w after it.
# ----------------------------------------------------------------------------------------------
def pnp_admm_cnc(y, mask, denoise, iter_num, alpha, lambda1, reo, b, denoise2=None, trace=()):
    """S6:256-308 (one model) / S6:485-525 (``denoise2`` given = the DnCNN pair variant)."""
    denoise2 = denoise2 or denoise
    x, z, w = init_state(y)
    rec = {}
    for i in range(iter_num):
        x = dc_step(z, w, y, mask, reo)                       # S6:266-271 (NumPy, host)
        x = np.float32(x)                                     # S6:273-275 single2tensor4().float()
        z = np.float32(np.absolute(z))                        # S6:277-280
        w = np.float32(np.absolute(w))                        # S6:282-285
        s = denoise(z, i)                                     # S6:300
        t = (1 - alpha) * z + alpha * (x + w) + alpha * reo * lambda1 * b * (z - s)   # S6:301
        t = np.float32(t)
        z = denoise2(t, i)                                    # S6:302
        w = w + x - z                                         # S6:305
        x = np.clip(x, 0, 1).astype(np.float32)               # S6:306
        z = np.clip(z, 0, 1).astype(np.float32)               # S6:307
        w = np.clip(w, 0, 1).astype(np.float32)               # S6:308
        if (i + 1) in trace:
            rec[i + 1] = (x.copy(), z.copy(), w.copy())
    return (x, rec) if trace else x


def pnp_admm_l1(y, mask, denoise, iter_num, reo, trace=()):
    """S3:249-296."""
    x, z, w = init_state(y)
    rec = {}
    for i in range(iter_num):
        x = dc_step(z, w, y, mask, reo)                       # S3:259-264
        x = np.float32(x)
        z = np.float32(np.absolute(z))                        # S3:266-276 (same marshalling)
        w = np.float32(np.absolute(w))
        z = denoise(np.float32(x + w), i)                     # S3:290
        w = w + x - z                                         # S3:293
        x = np.clip(x, 0, 1).astype(np.float32)               # S3:294-296
        z = np.clip(z, 0, 1).astype(np.float32)
        w = np.clip(w, 0, 1).astype(np.float32)
        if (i + 1) in trace:
            rec[i + 1] = (x.copy(), z.copy(), w.copy())
    return (x, rec) if trace else x


# ----------------------------------------------------------------------------------------------
# a13  sigma / rho schedules                                       utils/utils_pnp.py:14-34
# ----------------------------------------------------------------------------------------------
def get_rho_sigma(sigma=2.55 / 255, iter_num=15, modelSigma1=49.0, modelSigma2=2.55, w=1.0):
    modelSigmaS = np.logspace(np.log10(modelSigma1), np.log10(modelSigma2), iter_num).astype(np.float32)
    modelSigmaS_lin = np.linspace(modelSigma1, modelSigma2, iter_num).astype(np.float32)
    sigmas = (modelSigmaS * w + modelSigmaS_lin * (1 - w)) / 255.
    rhos = list(map(lambda x: 0.23 * (sigma ** 2) / (x ** 2), sigmas))
    return rhos, sigmas


def get_rho_sigma1(sigma=2.55 / 255, iter_num=15, modelSigma1=49.0, modelSigma2=2.55, lamda=3.0):
    modelSigmaS = np.logspace(np.log10(modelSigma1), np.log10(modelSigma2), iter_num).astype(np.float32)
    sigmas = modelSigmaS / 255.
    rhos = list(map(lambda x: (sigma ** 2) / (x ** 2) / lamda, sigmas))
    return rhos, sigmas


# ----------------------------------------------------------------------------------------------
# a14  Df = A^H (A x - y), plus the A / A^H it is made of            utils/utils.py:50-55
# ----------------------------------------------------------------------------------------------
def Df(x, mask, y):
    res = np.fft.fft2(x) * mask
    index = np.nonzero(mask)
    res[index] = res[index] - y[index]
    return np.fft.ifft2(res)


def A(x, mask):
    """Forward operator implied by S4:102: masked unnormalised 2-D DFT."""
    return np.fft.fft2(x) * mask


def AH(k, mask):
    """Adjoint in the reference's scaling convention (ifft2 carries 1/(HW); utils/utils.py:54)."""
    return np.fft.ifft2(k * mask)


# ----------------------------------------------------------------------------------------------
# a15  metrics                                               utils/utils_image.py:543-564, 622-636
# ----------------------------------------------------------------------------------------------
def calculate_psnr(img1, img2, border=0):
    h, w = img1.shape[:2]
    img1 = img1[border:h - border, border:w - border].astype(np.float64)
    img2 = img2[border:h - border, border:w - border].astype(np.float64)
    mse = np.mean((img1 - img2) ** 2)
    if mse == 0:
        return float('inf')
    return 20 * np.log10(255.0 / np.sqrt(mse))


def calculate_re(img1, img2, border=0):
    h, w = img1.shape[:2]
    img1 = img1[border:h - border, border:w - border].astype(np.float64)
    img2 = img2[border:h - border, border:w - border].astype(np.float64)
    return np.linalg.norm(img2 - img1) / np.linalg.norm(img2)


def psnr255(x, im_orig):
    """``util.psnr`` (max = 255, complex input allowed): the "zero-filling psnr" print, S4:104."""
    M, N = np.shape(x)
    mse = (np.sum((np.absolute(x - im_orig)) ** 2)) / (M * N)
    return 10 * np.log10(255 * 255 / mse)


def _gauss_kernel(n=11, sigma=1.5):
    g = np.exp(-((np.arange(n) - (n - 1) / 2.0) ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def calculate_ssim(img1, img2):
    """utils/utils_image.py:593-615: Gaussian 11/1.5 window, valid region (the [5:-5] crop makes
    cv2.filter2D's border mode irrelevant)."""
    from scipy.signal import correlate2d
    C1 = (0.01 * 255) ** 2
    C2 = (0.03 * 255) ** 2
    img1 = img1.astype(np.float64)
    img2 = img2.astype(np.float64)
    k = _gauss_kernel()
    window = np.outer(k, k)
    f = lambda a: correlate2d(a, window, mode='valid')
    mu1, mu2 = f(img1), f(img2)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 ** 2, mu2 ** 2, mu1 * mu2
    sigma1_sq = f(img1 ** 2) - mu1_sq
    sigma2_sq = f(img2 ** 2) - mu2_sq
    sigma12 = f(img1 * img2) - mu1_mu2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean()


# ----------------------------------------------------------------------------------------------
# Synthetic workload of SURVEY.md section 8(d) -- shared by tests and bench so CPU and GPU see
# identical inputs.  Nothing here comes from the reference except the mask it is handed.
# ----------------------------------------------------------------------------------------------
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
        img += amp * np.clip((1.0 - r) * 8.0, 0.0, 1.0)      # soft edge ~1/8 of the radius
    img = np.clip(img, 0, 1)
    return np.float32(np.round(img * 255.) / 255.)


def kspace_noise(b, H=256, W=256, std=15.0):
    rng = np.random.default_rng(777 + b)
    return std * (rng.standard_normal((H, W)) + 1j * rng.standard_normal((H, W)))


def synthetic_problem(b, mask, H=256, W=256):
    """-> (img float32 [H,W], y complex128 [H,W]); the GPU gets y cast to complex64."""
    img = phantom(b, H, W)
    y = synthesize(img, mask.astype(np.float64), kspace_noise(b, H, W))
    return img, y


def synthetic_mask(kind, H, W, rate=0.30, seed=4242):
    """Seeded masks for sizes the reference has no .mat for (cfg 5, 512x512).  Un-shifted FFT
    layout, DC sampled, ~``rate`` of the points: 'random' (variable density), 'radial'
    (lines through DC), 'cartesian' (full rows, dense centre)."""
    rng = np.random.default_rng(seed + {'random': 0, 'radial': 1, 'cartesian': 2}[kind])
    fy = np.fft.fftfreq(H)[:, None]
    fx = np.fft.fftfreq(W)[None, :]
    if kind == 'random':
        r = np.sqrt(fy ** 2 + fx ** 2) / 0.5
        pdf = (1 - np.clip(r, 0, 1)) ** 3 + 0.02
        lo, hi = 0.0, 50.0
        for _ in range(60):                       # bisection on the scale to hit the rate
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
