"""The oracle (oracle/admm_oracle.py) pinned against vectors produced by the unmodified
reference scripts (oracle/make_golden.py) and the scalar answers in the reference's own logs."""
import re

import numpy as np
import pytest

from oracle import admm_oracle as O
from conftest import rel_l2

MASKS = {'random30': 'Q_Random30', 'radial30': 'Q_Radial30', 'cartesian30': 'Q_Cartesian30'}


def _problem(golden_inputs, mname):
    img = O.requantise(golden_inputs['gray'])
    mask = golden_inputs['masks'][MASKS[mname]].astype(np.float64)
    y = O.synthesize(img, mask, golden_inputs['noises'])
    return img, mask, y


@pytest.mark.parametrize('n', [1, 2, 5, 10, 50])
def test_l1_matches_reference_trace(golden_inputs, golden_admm, n):
    _, mask, y = _problem(golden_inputs, 'random30')
    x = O.admm_l1(y, mask, iter_num=n)                      # S1:171 presets
    ref = golden_admm['l1_random30_it%d' % n]
    if ref.dtype == np.float64:
        assert rel_l2(x, ref) <= 1e-14                      # float64 fixture: bit-level
    else:
        assert np.array_equal(x.astype(np.float32), ref)    # float32 fixture: exact after cast


@pytest.mark.parametrize('n', [1, 2, 5, 10, 50])
def test_cnc_matches_reference_trace(golden_inputs, golden_admm, n):
    _, mask, y = _problem(golden_inputs, 'random30')
    x = O.admm_cnc(y, mask, iter_num=n)                     # S4:176 presets
    ref = golden_admm['cnc_random30_it%d' % n]
    if ref.dtype == np.float64:
        assert rel_l2(x, ref) <= 1e-13
    else:
        assert rel_l2(x, ref) <= 1e-7


@pytest.mark.parametrize('mname', ['radial30', 'cartesian30'])
def test_other_masks(golden_inputs, golden_admm, mname):
    _, mask, y = _problem(golden_inputs, mname)
    assert rel_l2(O.admm_l1(y, mask), golden_admm['l1_%s_it50' % mname]) <= 1e-7
    assert rel_l2(O.admm_cnc(y, mask), golden_admm['cnc_%s_it50' % mname]) <= 1e-7


def test_known_scalar_answers(golden_inputs, known_answers):
    """min / max / sum of out[0] and the log lines (PSNR/SSIM/RE) of the committed defaults.
    The authors' own logs say: L1 23.8683 dB / 0.5877 / 0.2028
    (results/Set1_dn_ADMM_L1/Set1_dn_ADMM_L1.log:284,287-288), CNC 24.5765 / 0.5600 / 0.1870
    (results/Set1_dn_ADMM_CNC/Set1_dn_ADMM_CNC.log:399-400)."""
    img, mask, y = _problem(golden_inputs, 'random30')
    gt = golden_inputs['gray']
    for tag, fn, authors in (('l1', O.admm_l1, (23.8683, 0.5877, 0.2028)),
                             ('cnc', O.admm_cnc, (24.5765, 0.5600, 0.1870))):
        x = fn(y, mask)
        ka = known_answers[tag]
        assert abs(x.sum() - ka['x_sum']) <= 1e-9 * ka['x_sum']
        assert abs(x.max() - ka['x_max']) <= 1e-12
        assert abs(x.min() - ka['x_min']) <= 1e-12
        psnr = O.calculate_psnr(x * 255, gt)
        ssim = O.calculate_ssim(x * 255, gt)
        re_ = O.calculate_re(x * 255, gt)
        got = [float(v) for v in re.findall(r'[-+]?\d+\.\d+', ka['log_line'].split(' - ')[1])]
        assert abs(psnr - got[0]) <= 0.006 and abs(ssim - got[1]) <= 6e-5 and abs(re_ - got[2]) <= 6e-5
        assert abs(psnr - authors[0]) <= 2e-4 and abs(ssim - authors[1]) <= 6e-5 and abs(re_ - authors[2]) <= 6e-5
    zf = O.psnr255(np.fft.ifft2(y) * 255, img * 255)
    assert ('zero-filling psnr = %.4f' % zf) in known_answers['l1']['stdout']


def test_sigma_schedule(golden_admm):
    rhos, sigmas = O.get_rho_sigma(sigma=max(0.255 / 255., 15 / 255.), iter_num=50,
                                   modelSigma1=49, modelSigma2=15, w=1.0)
    assert np.array_equal(np.asarray(sigmas), golden_admm['rho_sigma_sigmas'])
    assert np.array_equal(np.asarray(rhos, dtype=np.float64), golden_admm['rho_sigma_rhos'])
    rhos, sigmas = O.get_rho_sigma1(sigma=2.55 / 255, iter_num=15, modelSigma1=49.0, modelSigma2=2.55, lamda=3.0)
    assert np.array_equal(np.asarray(sigmas), golden_admm['rho_sigma1_sigmas'])
    assert np.array_equal(np.asarray(rhos, dtype=np.float64), golden_admm['rho_sigma1_rhos'])


def test_hermitian_half_spectrum_identity(golden_inputs):
    """The identity the fused kernels rely on: for real v,
    Re(ifft2(X)) == ifft2(V*(1-c*Mh) + c*Yh) with Mh=(m+flip m)/2, Yh=(m*y + conj flip(m*y))/2."""
    _, mask, y = _problem(golden_inputs, 'random30')
    rng = np.random.default_rng(0)
    z, w = rng.uniform(0, 1, (256, 256)), rng.uniform(-.1, .1, (256, 256))
    reo = 0.05
    x_ref = O.dc_step(z, w, y, mask, reo)
    flip = lambda a: np.roll(a[::-1, ::-1], (1, 1), (0, 1))
    Mh = (mask + flip(mask)) / 2
    Yh = (mask * y + np.conj(flip(mask * y))) / 2
    c = 1.0 / (1.0 + 1.0 / 2.0 / reo)
    X = np.fft.fft2(z - w) * (1 - c * Mh) + c * Yh
    x = np.abs(np.fft.ifft2(X).real)
    assert np.abs(np.fft.ifft2(X).imag).max() < 1e-12
    assert rel_l2(x, x_ref) < 1e-13
