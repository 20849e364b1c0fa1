"""Host logic that needs no GPU: the product's synthetic workload equals the oracle's copy."""
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
