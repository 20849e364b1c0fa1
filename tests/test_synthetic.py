"""Host logic that needs no GPU: the product's synthetic workload equals the oracle's copy."""
import os

import numpy as np

from oracle import admm_oracle as O
from pnp_admm_cnc_mri_amd import synthetic as S


def test_generators_agree():
    for b in (0, 7, 511):
        assert np.array_equal(S.phantom(b), O.phantom(b))
        assert np.array_equal(S.kspace_noise(b), O.kspace_noise(b))
    for kind in ('random', 'radial', 'cartesian'):
        m = S.synthetic_mask(kind, 512, 512)
        assert np.array_equal(m, O.synthetic_mask(kind, 512, 512))
        assert m[0, 0] == 1 and 0.25 < m.mean() < 0.35


def test_phantom_is_quantised_and_in_range():
    p = S.phantom(3)
    assert p.dtype == np.float32 and p.min() >= 0 and p.max() <= 1
    assert np.array_equal(np.round(p * 255) / 255, p.astype(np.float64).round(12)) or np.allclose(np.round(p * 255), p * 255, atol=1e-4)
    assert p.std() > 0.05


def test_reference_masks_fixture(golden_inputs):
    m = S.reference_masks()
    assert [int(m[k].sum()) for k in ('Q_Random30', 'Q_Radial30', 'Q_Cartesian30')] == [19674, 19294, 19456]
    assert all(m[k][0, 0] == 1 for k in m)


def test_host_metrics_and_imageio_match_the_oracle(golden_inputs, golden_admm, tmp_path):
    """pnp_admm_cnc_mri_amd.metrics / imageio (host utilities at the edges of the solvers) against the
    oracle's restatement of utils/utils_image.py."""
    from pnp_admm_cnc_mri_amd import metrics as M, imageio as IO
    gt = golden_inputs['gray']
    x = golden_admm['cnc_random30_it50']
    assert abs(M.calculate_psnr(x * 255, gt) - O.calculate_psnr(x * 255, gt)) <= 1e-12
    assert abs(M.calculate_re(x * 255, gt) - O.calculate_re(x * 255, gt)) <= 1e-15
    assert abs(M.calculate_ssim(x * 255, gt) - O.calculate_ssim(x * 255, gt)) <= 1e-12
    assert abs(M.psnr(x * 255, gt.astype(np.float64)) - O.psnr255(x * 255, gt.astype(np.float64))) <= 1e-12
    assert np.array_equal(IO.requantise(gt), O.requantise(gt))
    # gray decode: an RGB(A) PNG goes through OpenCV's fixed-point formula, an L PNG is taken as is
    from PIL import Image
    rgb = np.random.default_rng(0).integers(0, 256, (16, 24, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(tmp_path / 'c.png')
    Image.fromarray(rgb[..., 0]).save(tmp_path / 'g.png')
    r64 = rgb.astype(np.int64)
    want = ((4899 * r64[..., 0] + 9617 * r64[..., 1] + 1868 * r64[..., 2] + 8192) >> 14).astype(np.uint8)
    assert np.array_equal(IO.imread_gray(str(tmp_path / 'c.png')), want)
    assert np.array_equal(IO.imread_gray(str(tmp_path / 'g.png')), rgb[..., 0])
    assert IO.modcrop(np.zeros((21, 19)), 8).shape == (16, 16)
    assert [os.path.basename(p) for p in IO.get_image_paths(str(tmp_path))] == ['c.png', 'g.png']
