"""fp64 engine: the same device loops in double.  This is where the north star's 1e-5 / 0.01 dB bar is
met END TO END for the runs fp32 cannot hold (100 CNC iterations with the committed, locally
expansive presets): the reference itself computes in float64.  At 256x256 the loops run on the fused
"split chain" kernels in double (k_frows<double>, k_fcols2<double>); other shapes, and
set_fast_path(0), use the generic kernels in double."""
import numpy as np
import pytest

from oracle import admm_oracle as O
from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def P():
    import pnp_admm_cnc_mri_amd as P
    from pnp_admm_cnc_mri_amd import _lib
    assert _lib.device_count() >= 1
    return P


def test_config1_reference_inputs_f64(P, golden_inputs, golden_admm, known_answers):
    """05.png / noises.mat / Q_Random30, committed presets, 50 iterations: float64 fixtures of the
    unmodified reference, 1e-9 (the reference's first fft2 runs in complex64, S4:102 under
    NumPy 2; here y is the oracle's y, so everything after is double on both sides)."""
    img = O.requantise(golden_inputs['gray'])
    mask = golden_inputs['masks']['Q_Random30']
    y = O.synthesize(img, mask.astype(np.float64), golden_inputs['noises'])
    for fast, path in ((1, 'fused'), (0, 'generic')):
        with P.Engine(256, 256, Bmax=1, precision='f64') as eng:
            eng.set_fast_path(fast)
            eng.upload(y, mask)
            assert eng.path_name == path
            eng.init_state()
            eng.admm_l1(50, 0.1, 0.015)
            xl = eng.x()[0]
            eng.init_state()
            eng.admm_cnc(50, 0.45, 0.5, 0.05, 64)
            xc = eng.x()[0]
        assert xl.dtype == np.float64
        assert rel_l2(xl, golden_admm['l1_random30_it50']) <= 1e-9
        assert rel_l2(xc, golden_admm['cnc_random30_it50']) <= 1e-9
        assert abs(xl.sum() - known_answers['l1']['x_sum']) <= 1e-6 and abs(xc.sum() - known_answers['cnc']['x_sum']) <= 1e-6


@pytest.mark.parametrize('H', [256, 512])
def test_config2_100_cnc_iterations_f64(P, golden_inputs, H):
    """config 2's run length with the S4:176 presets (and config 5's 512x512 shape): <= 1e-5 end to
    end -- in fact ~1e-10, against 4e-4 for any float32 arithmetic."""
    if H == 256:
        masks = np.stack([golden_inputs['masks'][k] for k in ('Q_Random30', 'Q_Radial30', 'Q_Cartesian30')]).astype(np.uint8)
    else:
        masks = np.stack([O.synthetic_mask(k, H, H) for k in ('random', 'radial', 'cartesian')])
    B = 3
    ys = np.stack([O.synthetic_problem(b, masks[b], H, H)[1] for b in range(B)])
    mid = np.arange(B, dtype=np.int32)
    with P.Engine(H, H, Bmax=B, precision='f64') as eng:
        eng.upload(ys, masks, mid)
        eng.init_state()
        eng.admm_cnc(100, 0.45, 0.5, 0.05, 64)
        x = eng.x()
    for b in range(B):
        ref = O.admm_cnc(ys[b], masks[b], 100)
        assert rel_l2(x[b], ref) <= 1e-5
        assert rel_l2(x[b], ref) <= 1e-8, rel_l2(x[b], ref)


def test_entry_points_in_double_meet_the_north_star_bar(P, golden_inputs, golden_admm, known_answers, tmp_path):
    """Round-3 verdict, item 1: the reference's committed presets through the reference's signatures, in the
    reference's arithmetic.  ADMM_CNC(mask, noises, **opts) with precision='f64' synthesises y, initialises, iterates
    and measures in double on the device: 50 CNC iterations (S4:176) end <= 1e-5 from the fixture of the unmodified
    script (float32 ends at 2.6e-5: tests/test_gpu_parity.py), and PSNR / SSIM / RE are the authors' log lines."""
    mask = golden_inputs['masks']['Q_Random30'].astype(np.float64)
    gray = golden_inputs['gray'][None]
    out, info = P.ADMM_CNC(mask, golden_inputs['noises'], images=gray, results=str(tmp_path), return_info=True,
                           precision='f64', **P.PRESETS['ADMM_CNC'])
    assert len(out) == 22 and out[0].dtype == np.float64 and out[1].dtype == np.uint8
    err = rel_l2(out[0], golden_admm['cnc_random30_it50'])
    assert err <= 1e-5, err                                           # the north star's bar (measured ~2e-6: the first fft2, below)
    # authors' log (results/Set1_dn_ADMM_CNC/...log:399): 24.5765 / 0.5600 / 0.1870; the unmodified script under NumPy 2.2.6 here:
    import re as _re
    ps, ss, rr = map(float, _re.match(r'.*PSNR: ([\d.]+) dB; SSIM: ([\d.]+) ; RE: ([\d.]+)\.', known_answers['cnc']['log_line']).groups())
    assert abs(ps - 24.5765) <= 1.5e-4
    assert abs(info['psnr'][0] - ps) <= 1e-4 and abs(info['ssim'][0] - ss) <= 1e-4 and abs(info['re'][0] - rr) <= 1e-4
    out, info = P.ADMM_L1(mask, golden_inputs['noises'], images=gray, results=str(tmp_path), return_info=True,
                          precision='f64', **P.PRESETS['ADMM_L1'])
    assert rel_l2(out[0], golden_admm['l1_random30_it50']) <= 1e-6
    ps, ss, rr = map(float, _re.match(r'.*PSNR: ([\d.]+) dB; SSIM: ([\d.]+) ; RE: ([\d.]+)\.', known_answers['l1']['log_line']).groups())
    assert abs(ps - 23.8683) <= 6e-3                                  # S1 prints two decimals (S1:150): 23.87
    assert abs(info['psnr'][0] - 23.8683) <= 1e-4 and abs(info['ssim'][0] - 0.5877) <= 1e-4 and abs(info['re'][0] - 0.2028) <= 1e-4
    # y= instead of images=: the reference's own y (complex64 first transform, then promoted) -> the 1e-9 of the engine tests
    y = O.synthesize(O.requantise(golden_inputs['gray']), mask, golden_inputs['noises'])
    out = P.ADMM_CNC(mask, None, y=y[None], results=str(tmp_path), precision='f64', **P.PRESETS['ADMM_CNC'])
    assert rel_l2(out[0], golden_admm['cnc_random30_it50']) <= 1e-9
    with pytest.raises(ValueError):
        P.ADMM_CNC(mask, None, y=y[None], precision='f16')


@pytest.mark.parametrize('H', [256, 512])
def test_synthesis_metrics_and_ssim_in_double(P, golden_inputs, H):
    """pnp_synthesize_problem_f64 / pnp_metrics_f64 / pnp_ssim_f64 against the oracle: y differs from NumPy's by the
    float32 round-off of NumPy's own first transform (S4:102 runs it in complex64; here the image is widened and
    transformed in double), everything else agrees to double round-off."""
    from pnp_admm_cnc_mri_amd import synthetic as S
    B = 3
    if H == 256:
        masks = np.stack([golden_inputs['masks'][k] for k in ('Q_Random30', 'Q_Radial30', 'Q_Cartesian30')]).astype(np.uint8)
    else:
        masks = np.stack([O.synthetic_mask(k, H, H) for k in ('random', 'radial', 'cartesian')])
    img, noise = S.batch(40, B, H, H)
    noise = noise.astype(np.complex128) * (1 + 1e-9)                  # not representable in complex64: the double path must keep it
    mid = np.arange(B, dtype=np.int32)
    gt = np.clip(np.round(img * 255), 0, 255).astype(np.uint8)
    with P.Engine(H, H, Bmax=B, precision='f64') as eng:
        eng.synthesize(img, noise, masks, mid)
        y = eng.download_y()
        assert y.dtype == np.complex128
        for b in range(B):
            y_ref64 = np.fft.fft2(img[b].astype(np.float64)) * masks[b] + noise[b]       # all-double control
            assert rel_l2(y[b], y_ref64) <= 1e-14
            assert rel_l2(y[b], O.synthesize(img[b], masks[b].astype(np.float64), noise[b])) <= 5e-7   # NumPy's complex64 transform
        eng.synthesize(img, noise[0], masks, mid)                      # one shared noise array (the reference's noises.mat)
        assert rel_l2(eng.download_y()[2], np.fft.fft2(img[2].astype(np.float64)) * masks[2] + noise[0]) <= 1e-14
        eng.init_state()
        eng.admm_cnc(3, 0.45, 0.5, 0.05, 64)
        x = eng.x()
        psnr, re = eng.metrics(None, gt)
        ssim = eng.ssim(None, gt)
    for b in range(B):
        assert abs(psnr[b] - O.calculate_psnr(x[b] * 255, gt[b])) <= 1e-9
        assert abs(re[b] - O.calculate_re(x[b] * 255, gt[b])) <= 1e-12
        assert abs(ssim[b] - O.calculate_ssim(x[b] * 255, gt[b])) <= 1e-10


def test_f64_context_rejects_float_entry_points(P):
    from pnp_admm_cnc_mri_amd._lib import PnpError
    with P.Engine(256, 256, Bmax=1, precision='f64') as eng:
        eng.upload(np.zeros((1, 256, 256), np.complex128), np.ones((256, 256), np.uint8))
        assert eng.path_name == 'fused'
        with pytest.raises(PnpError):                                  # the PnP path's float32 operators stay float-only
            eng.dc_step(np.zeros(1, np.float32), np.zeros(1, np.float32), np.zeros(1, np.float32), 0.05)
    with P.Engine(256, 256, Bmax=1) as eng:                            # ... and the double entry points need a double context
        eng.upload(np.zeros((1, 256, 256), np.complex64), np.ones((256, 256), np.uint8))
        from pnp_admm_cnc_mri_amd import _lib
        buf = np.empty((1, 256, 256), np.complex128)
        with pytest.raises(PnpError):
            _lib.check(eng._L.pnp_download_y_f64(eng._ctx, buf.ctypes.data, 0))
    with P.Engine(512, 512, Bmax=1, precision='f64') as eng:
        eng.upload(np.zeros((1, 512, 512), np.complex128), np.ones((512, 512), np.uint8))
        assert eng.path_name == 'generic'


def test_config2_full_batch_on_the_fused_f64_path(P):
    """Config 2 as BASELINE.json states it -- 512 slices of 256x256, Q_Random30, S4:176 presets, 100 CNC
    iterations -- on the fused double-precision path: (a) every slice agrees with the generic double
    kernels (a different FFT factorisation and data flow) to 1e-9; (b) slices 0, 255 and 511 agree
    with the float64 oracle to 1e-8, three orders inside the north star's 1e-5; (c) a run split into two
    calls gives the same bits, an odd batch the same result."""
    from pnp_admm_cnc_mri_amd import synthetic as S
    B = 512
    mask = S.reference_masks()['Q_Random30'].astype(np.uint8)
    with P.Engine(256, 256, Bmax=B) as e32:                   # measurements synthesised on the device (float path)
        img, noise = S.batch(0, B)
        e32.synthesize(img, noise, mask)
        ys = e32.download_y().astype(np.complex128)
    res = {}
    for fast in (1, 0):
        with P.Engine(256, 256, Bmax=B, precision='f64') as eng:
            eng.set_fast_path(fast)
            eng.upload(ys, mask)
            assert eng.path_name == ('fused' if fast else 'generic')
            eng.init_state()
            eng.admm_cnc(100, 0.45, 0.5, 0.05, 64)
            res[fast] = eng.x()
            if fast:
                eng.init_state()
                eng.admm_cnc(37, 0.45, 0.5, 0.05, 64)
                eng.admm_cnc(63, 0.45, 0.5, 0.05, 64)
                assert np.array_equal(eng.x(), res[1])          # resumable: 37 + 63 == 100 iterations, bit for bit
    num = np.sqrt(((res[1] - res[0]) ** 2).sum(axis=(1, 2)))
    den = np.sqrt((res[0] ** 2).sum(axis=(1, 2)))
    assert (num / den).max() <= 1e-9, (num / den).max()      # measured 1.0e-10: 1e-16 rounding grown ~1.08x per iteration
    for b in (0, 255, 511):
        ref = O.admm_cnc(ys[b], mask, 100)
        assert rel_l2(res[1][b], ref) <= 1e-8, (b, rel_l2(res[1][b], ref))
    with P.Engine(256, 256, Bmax=7, precision='f64') as eng:   # odd batch: the last pair has one slice
        eng.upload(ys[:7], mask)
        eng.init_state()
        eng.admm_cnc(100, 0.45, 0.5, 0.05, 64)
        x7 = eng.x()
    assert np.array_equal(x7[:6], res[1][:6])                  # same pairs -> same bits
    assert rel_l2(x7[6], res[1][6]) <= 1e-9                    # slice 6 lost its partner in the shared complex transform: round-off only
