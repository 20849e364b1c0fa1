"""GPU parity tests: the HIP path (through the C ABI, ctypes) against the oracle and the golden
vectors from the unmodified reference.  Run on the MI355X box with `pytest -m gpu`.

Tolerances (fp32 device arithmetic vs the float64 oracle):
  * single transforms / operators : relative L2 <= 2e-6   (fp32 FFT round-off, 256..512 points);
    one x-update from identical state (teacher-forced): <= 5e-7 (measured 1.6e-7 .. 2.0e-7)
  * whole solves, north-star bar   : relative L2 <= 1e-5 and |dPSNR| <= 0.01 dB
    -- holds for ADMM_L1 at any iteration count and for ADMM_CNC up to ~40 iterations; the
    committed CNC defaults (reo*lambda*b = 1.6 > 1) make the iteration map locally expansive, so
    fp32 round-off grows ~1.08x per iteration in *any* fp32 implementation (NumPy's own
    float32 run: 1e-5 at 50 iterations, 4e-4 at 100).  Those cases are pinned
    per iteration (teacher-forced, <= 2e-6), by PSNR within 0.01 dB, and end to end against the
    NumPy-float32 precision control (test_cnc_100_iterations_config2).
"""
import os

import numpy as np
import pytest

from oracle import admm_oracle as O
from conftest import rel_l2

pytestmark = pytest.mark.gpu

MASKS = {'random30': 'Q_Random30', 'radial30': 'Q_Radial30', 'cartesian30': 'Q_Cartesian30'}


@pytest.fixture(scope='module')
def P():
    import pnp_admm_cnc_mri_amd as P
    from pnp_admm_cnc_mri_amd import _lib
    assert _lib.device_count() >= 1, 'no HIP device visible: GPU tests must not pass silently'
    return P


@pytest.fixture(scope='module')
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


PATHS = ('generic', 'fused', 'slice')


class _engine_on:
    """Engine forced onto one of the three kernel families: 'generic' (set_fast_path(0)), 'fused' (two-launch,
    PNP_SLICE=0) or 'slice' (the slice-resident kernel bench.py times, PNP_SLICE=1 so that it also takes a batch
    of one slice).  check() asserts pnp_path_name after the problem is uploaded."""

    def __init__(self, P, path, H=256, W=256, Bmax=1):
        self.P, self.path, self.args = P, path, (H, W, Bmax)

    def __enter__(self):
        self.saved = os.environ.get('PNP_SLICE')
        os.environ['PNP_SLICE'] = '1' if self.path == 'slice' else '0'
        try:
            self.eng = self.P.Engine(self.args[0], self.args[1], Bmax=self.args[2])
        finally:
            if self.saved is None:
                os.environ.pop('PNP_SLICE', None)
            else:
                os.environ['PNP_SLICE'] = self.saved
        self.eng.set_fast_path(0 if self.path == 'generic' else 1)
        self.eng.check = lambda: self._check()
        return self.eng

    def _check(self):
        assert self.eng.path_name == self.path, (self.eng.path_name, self.path)

    def __exit__(self, *exc):
        self.eng.close()


def _masks(golden_inputs):
    return np.stack([golden_inputs['masks'][MASKS[k]] for k in ('random30', 'radial30', 'cartesian30')]).astype(np.uint8)


def _synthetic(B, masks, H=256, W=256, mask_id=None):
    mask_id = np.arange(B) % len(masks) if mask_id is None else mask_id
    imgs, ys = [], []
    for b in range(B):
        img, y = O.synthetic_problem(b, masks[mask_id[b]], H, W)
        imgs.append(img)
        ys.append(y)
    return np.stack(imgs), np.stack(ys), mask_id.astype(np.int32)


# ------------------------------------------------------------------------------------------------
# transforms and operators
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('H,W', [(256, 256), (512, 512), (256, 512), (512, 256)])
def test_fft2_roundtrip_and_numpy(P, torch, H, W):
    rng = np.random.default_rng(1)
    B = 3
    a = (rng.standard_normal((B, H, W)) + 1j * rng.standard_normal((B, H, W))).astype(np.complex64)
    with P.Engine(H, W, Bmax=B) as eng:
        t = torch.from_numpy(a).cuda()
        k = P.utils_pnp.fft2(eng, t)
        assert rel_l2(k.cpu().numpy(), np.fft.fft2(a.astype(np.complex128))) <= 2e-6
        back = P.utils_pnp.ifft2(eng, k)
        assert rel_l2(back.cpu().numpy(), a) <= 2e-6
        ki = P.utils_pnp.ifft2(eng, t)
        assert rel_l2(ki.cpu().numpy(), np.fft.ifft2(a.astype(np.complex128))) <= 2e-6


def test_fft2_linearity_and_parseval(P, torch):
    rng = np.random.default_rng(2)
    a = (rng.standard_normal((2, 256, 256)) + 1j * rng.standard_normal((2, 256, 256))).astype(np.complex64)
    with P.Engine(256, 256, Bmax=2) as eng:
        ta = torch.from_numpy(a).cuda()
        k = P.utils_pnp.fft2(eng, ta)
        e_x = float((ta.abs() ** 2).sum())
        e_k = float((k.abs() ** 2).sum()) / 65536
        assert abs(e_x - e_k) <= 1e-5 * e_x
        k2 = P.utils_pnp.fft2(eng, 2.5 * ta)
        assert rel_l2(k2.cpu().numpy(), 2.5 * k.cpu().numpy()) <= 1e-6
        delta = torch.zeros((1, 256, 256), dtype=torch.complex64, device='cuda')
        delta[0, 0, 0] = 1
        assert torch.allclose(P.utils_pnp.fft2(eng, delta), torch.ones_like(delta))


def test_operators_A_AH_Df(P, torch, golden_inputs):
    masks = _masks(golden_inputs)
    B = 3
    imgs, ys, mid = _synthetic(B, masks)
    with P.Engine(256, 256, Bmax=B) as eng:
        eng.upload(ys, masks, mid)
        x = torch.from_numpy(imgs).cuda()
        k = P.utils_pnp.A(eng, x).cpu().numpy()
        for b in range(B):
            assert rel_l2(k[b], O.A(imgs[b].astype(np.float64), masks[mid[b]])) <= 2e-6
        kk = torch.from_numpy(ys.astype(np.complex64)).cuda()
        ah = P.utils_pnp.AH(eng, kk).cpu().numpy()
        for b in range(B):
            assert rel_l2(ah[b], O.AH(ys[b], masks[mid[b]])) <= 2e-6
        df = P.utils_pnp.Df(eng, x).cpu().numpy()
        for b in range(B):
            # the device holds y as complex64: compare against the oracle on the same y
            ref = O.Df(imgs[b].astype(np.float64), masks[mid[b]].astype(np.float64), ys[b].astype(np.complex64).astype(np.complex128))
            assert rel_l2(df[b], ref) <= 5e-6
        # adjointness <A x, k> == N <x, A^H k>  (ifft2 carries 1/N)
        lhs = np.vdot(k[0], ys[0])
        rhs = 65536 * np.vdot(imgs[0].astype(np.complex128), ah[0])
        assert abs(lhs - rhs) <= 1e-4 * abs(lhs)


def test_synthesize_and_init(P, golden_inputs):
    """y = fft2(img)*mask + noises and z0 = |ifft2(y)| on the reference's real inputs (S4:101-109)."""
    img = O.requantise(golden_inputs['gray'])
    mask = golden_inputs['masks']['Q_Random30']
    y_ref = O.synthesize(img, mask.astype(np.float64), golden_inputs['noises'])
    with P.Engine(256, 256, Bmax=1) as eng:
        eng.synthesize(img, golden_inputs['noises'], mask)
        y = eng.download_y()[0]
        assert rel_l2(y, y_ref) <= 2e-6
        eng.init_state()
        z, w = eng.get_state()
        x0, _, _ = O.init_state(y_ref)
        assert rel_l2(z[0], x0) <= 2e-6
        assert not w.any()


@pytest.mark.parametrize('H,W', [(256, 256), (512, 512)])
def test_dc_step_teacher_forced(P, torch, golden_inputs, H, W):
    """One x-update from identical (z, w): the per-iteration parity that bounds everything else."""
    if H == 256:
        masks = _masks(golden_inputs)
    else:
        masks = np.stack([O.synthetic_mask(k, H, W) for k in ('random', 'radial', 'cartesian')])
    B = 4
    imgs, ys, mid = _synthetic(B, masks, H, W)
    rng = np.random.default_rng(5)
    z = rng.uniform(0, 1, (B, H, W)).astype(np.float32)
    w = rng.uniform(-0.1, 0.1, (B, H, W)).astype(np.float32)
    for fast in (0, 1):
        with P.Engine(H, W, Bmax=B) as eng:
            eng.set_fast_path(fast)
            eng.upload(ys, masks, mid)
            x = P.utils_pnp.dc_solve(eng, torch.from_numpy(z).cuda(), torch.from_numpy(w).cuda(), 0.05).cpu().numpy()
            for b in range(B):
                ref = O.dc_step(z[b].astype(np.float64), w[b].astype(np.float64),
                                ys[b].astype(np.complex64).astype(np.complex128), masks[mid[b]], 0.05)
                assert rel_l2(x[b], ref) <= 5e-7, (fast, b)          # measured 1.6e-7 .. 2.0e-7


def test_prox_kernels(P, torch):
    rng = np.random.default_rng(7)
    B = 2
    x = rng.uniform(0, 1, (B, 256, 256)).astype(np.float32)
    z = rng.uniform(-0.05, 1, (B, 256, 256)).astype(np.float32)
    w = rng.uniform(-0.2, 0.2, (B, 256, 256)).astype(np.float32)
    masks = np.ones((1, 256, 256), np.uint8)
    with P.Engine(256, 256, Bmax=B) as eng:
        eng.upload(np.zeros((B, 256, 256), np.complex64), masks)
        tx, tz, tw = (torch.from_numpy(a.copy()).cuda() for a in (x, z, w))
        P.utils_pnp.prox_l1(eng, tx, tz, tw, 0.0015)
        zr, wr = O.l1_step(x.astype(np.float64), z.astype(np.float64), w.astype(np.float64), 0.1, 0.015)
        assert np.abs(tz.cpu().numpy() - zr).max() <= 2e-7 and np.abs(tw.cpu().numpy() - wr).max() <= 2e-7
        tx, tz, tw = (torch.from_numpy(a.copy()).cuda() for a in (x, z, w))
        P.utils_pnp.prox_cnc(eng, tx, tz, tw, 0.45, 0.5, 0.05, 64)
        zr, wr = O.cnc_step(x.astype(np.float64), z.astype(np.float64), w.astype(np.float64), 0.45, 0.5, 0.05, 64)
        assert np.abs(tz.cpu().numpy() - zr).max() <= 5e-7 and np.abs(tw.cpu().numpy() - wr).max() <= 5e-7


# ------------------------------------------------------------------------------------------------
# whole solves against the golden vectors of the unmodified reference (config 1)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('path', PATHS)
@pytest.mark.parametrize('n', [1, 2, 5, 10, 50])
def test_admm_l1_golden(P, golden_inputs, golden_admm, n, path):
    img = O.requantise(golden_inputs['gray'])
    mask = golden_inputs['masks']['Q_Random30']
    with _engine_on(P, path) as eng:
        eng.synthesize(img, golden_inputs['noises'], mask)
        eng.check()
        eng.init_state()
        eng.admm_l1(n, 0.1, 0.015)                     # S1:171 presets
        x = eng.x()[0]
    ref = golden_admm['l1_random30_it%d' % n]
    assert rel_l2(x, ref) <= 1e-5
    gt = golden_inputs['gray']
    assert abs(O.calculate_psnr(x.astype(np.float64) * 255, gt) - O.calculate_psnr(ref.astype(np.float64) * 255, gt)) <= 0.01


@pytest.mark.parametrize('path', PATHS)
@pytest.mark.parametrize('n', [1, 2, 5, 10, 50])
def test_admm_cnc_golden(P, golden_inputs, golden_admm, n, path):
    img = O.requantise(golden_inputs['gray'])
    mask = golden_inputs['masks']['Q_Random30']
    with _engine_on(P, path) as eng:
        eng.synthesize(img, golden_inputs['noises'], mask)
        eng.check()
        eng.init_state()
        eng.admm_cnc(n, 0.45, 0.5, 0.05, 64)           # S4:176 presets (expansive: reo*lambda*b = 1.6)
        x = eng.x()[0]
    ref = golden_admm['cnc_random30_it%d' % n]
    tol = 1e-5 if n <= 10 else 1e-4                    # see module docstring
    assert rel_l2(x, ref) <= tol
    gt = golden_inputs['gray']
    assert abs(O.calculate_psnr(x.astype(np.float64) * 255, gt) - O.calculate_psnr(ref.astype(np.float64) * 255, gt)) <= 0.01


def test_solver_entry_points_on_reference_inputs(P, golden_inputs, golden_admm, known_answers, tmp_path):
    """ADMM_L1 / ADMM_CNC called the way the reference's __main__ calls them (S1:194, S4:202),
    images handed in instead of read from testsets/Set1; metrics must reproduce the log lines."""
    mask = golden_inputs['masks']['Q_Random30'].astype(np.float64)
    gray = golden_inputs['gray'][None]
    out, info = P.ADMM_L1(mask, golden_inputs['noises'], images=gray, results=str(tmp_path), return_info=True,
                          **P.PRESETS['ADMM_L1'])
    assert len(out) == 22 and out[0].dtype == np.float64 and out[1].dtype == np.uint8
    assert rel_l2(out[0], golden_admm['l1_random30_it50']) <= 1e-5
    assert abs(info['psnr'][0] - 23.8683) <= 0.01 and abs(info['re'][0] - 0.2028) <= 1e-4
    out, info = P.ADMM_CNC(mask, golden_inputs['noises'], images=gray, results=str(tmp_path), return_info=True,
                           **P.PRESETS['ADMM_CNC'])
    assert rel_l2(out[0], golden_admm['cnc_random30_it50']) <= 1e-4
    assert abs(info['psnr'][0] - 24.5765) <= 0.01 and abs(info['re'][0] - 0.1870) <= 1e-4


@pytest.mark.parametrize('path', PATHS)
@pytest.mark.parametrize('mname', ['radial30', 'cartesian30'])
def test_other_masks_golden(P, golden_inputs, golden_admm, mname, path):
    img = O.requantise(golden_inputs['gray'])
    mask = golden_inputs['masks'][MASKS[mname]]
    with _engine_on(P, path) as eng:
        eng.synthesize(img, golden_inputs['noises'], mask)
        eng.check()
        eng.init_state()
        eng.admm_l1(50, 0.1, 0.015)
        assert rel_l2(eng.x()[0], golden_admm['l1_%s_it50' % mname]) <= 1e-5


# ------------------------------------------------------------------------------------------------
# batched synthetic workload (config 2 shape, small B) vs the oracle, mixed masks, odd B
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('path', PATHS)
@pytest.mark.parametrize('B', [1, 5, 8])
def test_batched_mixed_masks_vs_oracle(P, golden_inputs, B, path):
    masks = _masks(golden_inputs)
    imgs, ys, mid = _synthetic(B, masks)
    with _engine_on(P, path, Bmax=8) as eng:
        eng.upload(ys, masks, mid)
        eng.check()
        eng.init_state()
        eng.admm_l1(100, 0.1, 0.015)
        xl1 = eng.x()
        eng.init_state()
        eng.admm_cnc(30, 0.45, 0.5, 0.05, 64)
        xcnc = eng.x()
    for b in range(B):
        y64 = ys[b].astype(np.complex64).astype(np.complex128)
        assert rel_l2(xl1[b], O.admm_l1(y64, masks[mid[b]], 100, 0.1, 0.015)) <= 1e-5, b
        assert rel_l2(xcnc[b], O.admm_cnc(y64, masks[mid[b]], 30, 0.45, 0.5, 0.05, 64)) <= 1e-5, b


@pytest.mark.parametrize('path', PATHS)
def test_cnc_100_iterations_config2(P, golden_inputs, path):
    """Config 2's iteration count (100) with the committed CNC presets.  The map is locally
    expansive, so fp32 round-off grows ~1.08x per iteration in any fp32 arithmetic: NumPy's own
    complex64/float32 run of the same lines ends 4e-4 away from the float64 reference.  Pinned:
      (a) teacher-forced: from the oracle's (z, w) at iterations 24, 49, 74, 98 one device
          iteration reproduces the oracle's next (x, z, w) to <= 2e-6 -- every step is right;
      (b) end to end: deviation from float64 <= 3x the deviation of the NumPy-float32 control,
          and PSNR against the ground truth within 0.01 dB of the reference's."""
    masks = _masks(golden_inputs)[:1]
    B = 2
    imgs, ys, mid = _synthetic(B, masks, mask_id=np.zeros(B, int))
    y64 = [ys[b].astype(np.complex64).astype(np.complex128) for b in range(B)]
    marks = (24, 49, 74, 98, 99)
    traces = [O.admm_cnc(y64[b], masks[0], 100, trace=marks + (25, 50, 75, 100))[1] for b in range(B)]
    with _engine_on(P, path, Bmax=B) as eng:
        eng.upload(ys, masks, mid)
        eng.check()
        for it in (24, 49, 74, 98):                                        # (a)
            z = np.stack([traces[b][it][1] for b in range(B)]).astype(np.float32)
            w = np.stack([traces[b][it][2] for b in range(B)]).astype(np.float32)
            eng.set_state(z, w)
            eng.admm_cnc(1, 0.45, 0.5, 0.05, 64)
            x1 = eng.x()
            z1, w1 = eng.get_state()
            for b in range(B):
                xr, zr, wr = O.dc_step(z[b].astype(np.float64), w[b].astype(np.float64), y64[b], masks[0], 0.05), None, None
                zr, wr = O.cnc_step(xr, z[b].astype(np.float64), w[b].astype(np.float64), 0.45, 0.5, 0.05, 64)
                assert rel_l2(x1[b], xr) <= 5e-7 and rel_l2(z1[b], zr) <= 2e-6
                assert np.abs(w1[b] - wr).max() <= 2e-6
        eng.init_state()                                                   # (b)
        eng.admm_cnc(100, 0.45, 0.5, 0.05, 64)
        x = eng.x()
    for b in range(B):
        ref = traces[b][100][0]
        control = rel_l2(O.admm_cnc_f32(y64[b], masks[0], 100), ref)
        assert rel_l2(x[b], ref) <= 3 * control, (rel_l2(x[b], ref), control)
        gt = np.uint8(np.round(imgs[b] * 255))
        assert abs(O.calculate_psnr(x[b].astype(np.float64) * 255, gt) - O.calculate_psnr(ref * 255, gt)) <= 0.01


def test_512_mixed_masks_vs_oracle(P):
    """config 5 shape (512x512, mask bank of 3, mask_id = b % 3), small batch."""
    H = W = 512
    masks = np.stack([O.synthetic_mask(k, H, W) for k in ('random', 'radial', 'cartesian')])
    assert all(0.25 < m.mean() < 0.35 for m in masks)
    B = 3
    imgs, ys, mid = _synthetic(B, masks, H, W)
    with P.Engine(H, W, Bmax=B) as eng:
        eng.upload(ys, masks, mid)
        eng.init_state()
        eng.admm_cnc(10, 0.45, 0.5, 0.05, 64)
        x = eng.x()
        eng.init_state()
        eng.admm_l1(30, 0.1, 0.015)
        xl = eng.x()
    for b in range(B):
        y64 = ys[b].astype(np.complex64).astype(np.complex128)
        assert rel_l2(x[b], O.admm_cnc(y64, masks[mid[b]], 10, 0.45, 0.5, 0.05, 64)) <= 1e-5
        assert rel_l2(xl[b], O.admm_l1(y64, masks[mid[b]], 30, 0.1, 0.015)) <= 1e-5


def test_config5_at_its_per_gpu_size(P):
    """BASELINE.json configs[4] as one GPU of the 8 sees it: 256 slices of 512x512, mask bank of 3 (mask_id = b % 3), the
    default chunked schedule (16-slice chunks round-robin over 4 HIP queues, `plan`), 6 CNC iterations: first, middle and
    last slice against the oracle <= 1e-5, the whole batch bit-identical to the unchunked schedule (chunk = -1)."""
    from pnp_admm_cnc_mri_amd import synthetic as S
    H = W = 512
    B = 256
    masks = np.stack([S.synthetic_mask(k, H, W) for k in ('random', 'radial', 'cartesian')])
    mid = (np.arange(B) % 3).astype(np.int32)
    img, noise = S.batch(7000, B, H, W)
    res = {}
    with P.Engine(H, W, Bmax=B) as eng:
        eng.synthesize(img, noise, masks, mid)
        assert eng.path_name == 'fused'
        assert eng.plan == {'queues': 4, 'chunk': 16, 'launches_per_iteration': 32}
        y_all = eng.download_y()
        for chunk in (0, -1):
            eng.set_schedule(queues=2, mixed_launches=False, chunk=chunk)
            eng.init_state()
            eng.admm_cnc(6, 0.45, 0.5, 0.05, 64)
            res[chunk] = eng.x().copy()
        eng.set_schedule(queues=2, mixed_launches=False, chunk=-1)
        assert eng.plan['chunk'] == B and eng.plan['queues'] == 1
    assert np.array_equal(res[0], res[-1])                                      # scheduling never changes a bit
    assert np.isfinite(res[0]).all()
    for b in (0, B // 2 - 1, B - 1):
        ref = O.admm_cnc(y_all[b].astype(np.complex128), masks[mid[b]], 6, 0.45, 0.5, 0.05, 64)
        assert rel_l2(res[0][b], ref) <= 1e-5, (b, rel_l2(res[0][b], ref))


def test_run_is_resumable_and_deterministic(P, golden_inputs):
    """10 + 15 iterations == 25 iterations bit for bit; two runs agree bit for bit."""
    masks = _masks(golden_inputs)
    imgs, ys, mid = _synthetic(4, masks)
    outs = []
    with P.Engine(256, 256, Bmax=4) as eng:
        eng.upload(ys, masks, mid)
        for split in ((25,), (10, 15), (25,)):
            eng.init_state()
            for n in split:
                eng.admm_cnc(n, 0.45, 0.5, 0.05, 64)
            outs.append(eng.x().copy())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_error_behaviour(P):
    from pnp_admm_cnc_mri_amd._lib import PnpError
    with pytest.raises(PnpError):
        P.Engine(100, 256)
    with P.Engine(256, 256, Bmax=2) as eng:
        with pytest.raises(PnpError):
            eng.init_state()                            # nothing uploaded
        with pytest.raises(PnpError):
            eng.upload(np.zeros((3, 256, 256), np.complex64), np.ones((256, 256), np.uint8))   # B > Bmax
        eng.upload(np.zeros((2, 256, 256), np.complex64), np.ones((256, 256), np.uint8))
        with pytest.raises(PnpError):
            eng.x()                                     # no iteration yet
        with pytest.raises(PnpError):
            eng.admm_l1(1, 0.1, 0.0)                    # reo must be > 0
        # thresholds must be >= 0: the fast paths evaluate soft(a, c) as a - med3(a, -c, c), which is the reference's
        # fmax(|a| - c, 0) * sign(a) (S1:18-19) for c >= 0 only -- rejected, not silently different
        eng.init_state()
        for bad in (lambda: eng.admm_l1(1, -0.1, 0.015), lambda: eng.admm_cnc(1, -0.45, 0.5, 0.05, 64),
                    lambda: eng.admm_cnc(1, 0.45, -0.5, 0.05, 64), lambda: eng.admm_cnc(1, 0.45, 0.5, 0.05, 0.0)):
            with pytest.raises(PnpError):
                bad()
        eng.admm_l1(1, 0.0, 0.015)                      # a zero threshold is fine: soft(a, 0) = a
        assert np.isfinite(eng.x()).all()


# ------------------------------------------------------------------------------------------------
# BASELINE.json's full size (config 2: 512 slices of 256x256, CNC presets) through
# size-independent properties: slices are independent, so any sub-batch must reproduce the same
# slices bit for bit, a permutation of the batch permutes the output, and x >= 0 is finite.
# ------------------------------------------------------------------------------------------------
def test_full_size_batch_properties(P, golden_inputs):
    from pnp_admm_cnc_mri_amd import synthetic as S
    masks = _masks(golden_inputs)
    B, K = 512, 25
    img, noise = S.batch(0, B)
    mid = (np.arange(B) % 3).astype(np.int32)
    with P.Engine(256, 256, Bmax=B) as eng:
        eng.synthesize(img, noise, masks, mid)
        assert eng.path_name == 'slice'                 # 512 slices = two full rounds of one workgroup per compute unit
        y = eng.download_y()
        eng.init_state()
        eng.admm_cnc(K, 0.45, 0.5, 0.05, 64)
        x = eng.x()
        # (1) permuted batch -> permuted output (pairs of slices share a complex transform, so this
        #     also shows that the pairing never leaks one slice into its partner beyond round-off:
        #     partners change under the permutation, results may move by fp32 round-off only)
        perm = np.random.default_rng(0).permutation(B)
        eng.upload(y[perm], masks, mid[perm])
        eng.init_state()
        eng.admm_cnc(K, 0.45, 0.5, 0.05, 64)
        xp = eng.x()
    assert np.isfinite(x).all() and x.min() >= 0
    sel = [0, 1, 255, 256, 511]
    for b in sel:
        pos = int(np.flatnonzero(perm == b)[0])
        assert rel_l2(xp[pos], x[b]) <= 2e-5, b
    # (2) the same slices in a batch of 6: bit-identical on the same path (the full batch runs slice-resident,
    #     one workgroup per slice, so force that path for the small batch too), and equal to round-off on the
    #     two-launch path the library would pick for 6 slices (different data flow, 25 CNC iterations)
    sub = [0, 1, 254, 255, 510, 511]
    res = {}
    for mode in ('1', '0'):
        os.environ['PNP_SLICE'] = mode
        try:
            with P.Engine(256, 256, Bmax=6) as eng:
                eng.upload(y[sub], masks, mid[sub])
                assert eng.path_name == ('slice' if mode == '1' else 'fused')
                eng.init_state()
                eng.admm_cnc(K, 0.45, 0.5, 0.05, 64)
                res[mode] = eng.x()
        finally:
            del os.environ['PNP_SLICE']
    assert np.array_equal(res['1'], x[sub])
    for k, b in enumerate(sub):
        assert rel_l2(res['0'][k], x[b]) <= 2e-5, b
    # (3) spot-check against the oracle
    for b in (0, 511):
        ref = O.admm_cnc(y[b].astype(np.complex128), masks[mid[b]], K)
        assert rel_l2(x[b], ref) <= 1e-5, b


def test_odd_batch_and_padding_slice(P, golden_inputs):
    """B odd: the last slice has no partner in its complex transform; its result must equal the
    result it gets with a partner present (round-off) and the oracle."""
    masks = _masks(golden_inputs)
    imgs, ys, mid = _synthetic(4, masks)
    with P.Engine(256, 256, Bmax=4) as eng:
        eng.upload(ys[:3], masks, mid[:3])
        eng.init_state()
        eng.admm_l1(20, 0.1, 0.015)
        x3 = eng.x()
        eng.upload(ys, masks, mid)
        eng.init_state()
        eng.admm_l1(20, 0.1, 0.015)
        x4 = eng.x()
    assert np.array_equal(x3[:2], x4[:2])
    assert rel_l2(x3[2], x4[2]) <= 2e-6
    assert rel_l2(x3[2], O.admm_l1(ys[2].astype(np.complex64).astype(np.complex128), masks[mid[2]], 20)) <= 1e-5


def test_l1_single_state_form_is_bit_identical(P, golden_inputs):
    """ADMM_L1's fused loop keeps only u = x + w between iterations (z = soft(u), w = u - z); the
    PNP_FUSED_L1_TWO_STATE=1 hook keeps z and w.  Same bits, and the state handed back is the
    genuine (z, w) pair."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import pnp_admm_cnc_mri_amd as P\n"
        "from pnp_admm_cnc_mri_amd import synthetic as S\n"
        "m = S.reference_masks(); masks = np.stack([m['Q_Random30'], m['Q_Radial30']]).astype(np.uint8)\n"
        "img, noise = S.batch(0, 5)\n"
        "eng = P.Engine(256, 256, Bmax=5); eng.synthesize(img, noise, masks, np.arange(5) %% 2); eng.init_state()\n"
        "eng.admm_l1(7, 0.1, 0.015); x1 = eng.x(); z1, w1 = eng.get_state()\n"
        "eng.admm_l1(1, 0.1, 0.015); eng.admm_l1(4, 0.1, 0.015); x2 = eng.x(); z2, w2 = eng.get_state()\n"
        "np.savez(sys.argv[1], x1=x1, z1=z1, w1=w1, x2=x2, z2=z2, w2=w2)\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outs = []
    for flag in ('0', '1'):
        path = '/tmp/l1_state_%s.npz' % flag
        subprocess.check_call([sys.executable, '-c', code, path], env=dict(os.environ, PNP_FUSED_L1_TWO_STATE=flag))
        outs.append(np.load(path))
    for k in ('x1', 'z1', 'w1', 'x2', 'z2', 'w2'):
        assert np.array_equal(outs[0][k], outs[1][k]), k
    # z = soft(u), w = u - z with u = z + w: w is clipped to the threshold
    assert np.abs(outs[0]['w2']).max() <= 0.1 * 0.015 + 1e-6     # u - fl(u - thr): thr up to an ulp of u


# ------------------------------------------------------------------------------------------------
# edge cases of the boundary
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('fast', [0, 1])
def test_degenerate_masks(P, torch, fast):
    """all-zero mask: the k-space blend is the identity, so x = |z - w| exactly up to FFT round-off;
    all-one mask: every point is blended; both against the oracle."""
    rng = np.random.default_rng(3)
    B = 2
    y = (rng.standard_normal((B, 256, 256)) + 1j * rng.standard_normal((B, 256, 256))).astype(np.complex64) * 50
    z = rng.uniform(0, 1, (B, 256, 256)).astype(np.float32)
    w = rng.uniform(-0.2, 0.2, (B, 256, 256)).astype(np.float32)
    masks = np.stack([np.zeros((256, 256), np.uint8), np.ones((256, 256), np.uint8)])
    mid = np.array([0, 1], np.int32)
    with P.Engine(256, 256, Bmax=B) as eng:
        eng.set_fast_path(fast)
        eng.upload(y, masks, mid)
        x = P.utils_pnp.dc_solve(eng, torch.from_numpy(z).cuda(), torch.from_numpy(w).cuda(), 0.3).cpu().numpy()
    assert np.abs(x[0] - np.abs(z[0] - w[0])).max() <= 2e-6
    ref = O.dc_step(z[1].astype(np.float64), w[1].astype(np.float64), y[1].astype(np.complex128), masks[1], 0.3)
    assert rel_l2(x[1], ref) <= 2e-6


def test_context_reuse_across_problems(P, golden_inputs):
    """One ctx, several uploads with different B / masks / solvers; zero iterations is a no-op."""
    masks = _masks(golden_inputs)
    imgs, ys, mid = _synthetic(6, masks)
    with P.Engine(256, 256, Bmax=6) as eng:
        for B in (6, 1, 4):
            eng.upload(ys[:B], masks, mid[:B])
            eng.init_state()
            z0, w0 = eng.get_state()
            eng.admm_l1(0, 0.1, 0.015)
            z1, w1 = eng.get_state()
            assert np.array_equal(z0, z1) and np.array_equal(w0, w1)
            eng.admm_l1(3, 0.1, 0.015)
            x = eng.x()
            assert x.shape == (B, 256, 256)
            for b in range(B):
                ref = O.admm_l1(ys[b].astype(np.complex64).astype(np.complex128), masks[mid[b]], 3)
                assert rel_l2(x[b], ref) <= 2e-6
        # a single mask for all slices (mask_id = None) and a 2-D mask argument
        eng.upload(ys[:2], masks[1])
        eng.init_state()
        eng.admm_cnc(2, 0.45, 0.5, 0.05, 64)
        ref = O.admm_cnc(ys[1].astype(np.complex64).astype(np.complex128), masks[1], 2)
        assert rel_l2(eng.x()[1], ref) <= 2e-6


def test_prox_edge_values(P, torch):
    """thresholds at and around the kinks: exact zeros, +-threshold, +-1/b, negative inputs."""
    B = 1
    vals = np.array([0.0, 1e-9, -1e-9, 0.0015, -0.0015, 0.00150001, 1 / 64, -1 / 64, 0.5, -0.5, 1.0, 2.0], np.float32)
    x = np.zeros((B, 256, 256), np.float32)
    z = np.zeros_like(x)
    w = np.zeros_like(x)
    n = len(vals)
    grid = np.stack(np.meshgrid(vals, vals, vals, indexing='ij'), -1).reshape(-1, 3)
    x.reshape(-1)[:len(grid)] = np.abs(grid[:, 0])
    z.reshape(-1)[:len(grid)] = grid[:, 1]
    w.reshape(-1)[:len(grid)] = grid[:, 2]
    with P.Engine(256, 256, Bmax=B) as eng:
        eng.upload(np.zeros((B, 256, 256), np.complex64), np.ones((256, 256), np.uint8))
        tx, tz, tw = (torch.from_numpy(a.copy()).cuda() for a in (x, z, w))
        P.utils_pnp.prox_l1(eng, tx, tz, tw, 0.0015)
        zr, wr = O.l1_step(x.astype(np.float64), z.astype(np.float64), w.astype(np.float64), 0.1, 0.015)
        assert np.abs(tz.cpu().numpy() - zr).max() <= 3e-7 and np.abs(tw.cpu().numpy() - wr).max() <= 3e-7
        tx, tz, tw = (torch.from_numpy(a.copy()).cuda() for a in (x, z, w))
        P.utils_pnp.prox_cnc(eng, tx, tz, tw, 0.45, 0.5, 0.05, 64)
        zr, wr = O.cnc_step(x.astype(np.float64), z.astype(np.float64), w.astype(np.float64), 0.45, 0.5, 0.05, 64)
        assert np.abs(tz.cpu().numpy() - zr).max() <= 1e-6 and np.abs(tw.cpu().numpy() - wr).max() <= 1e-6


def test_schedules_are_bit_identical(P):
    """Two-launch path (PNP_SLICE=0; a batch of 131 would otherwise run slice-resident): the batch may be
    split over HIP queues (PNP_FUSED_STREAMS) and row/column workgroups of different halves may share one
    launch (PNP_FUSED_SCHED=1): scheduling only -- every slice sees the same arithmetic, so all schedules
    give the same bits."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import pnp_admm_cnc_mri_amd as P\n"
        "from pnp_admm_cnc_mri_amd import synthetic as S\n"
        "m = S.reference_masks(); masks = np.stack([m['Q_Random30'], m['Q_Cartesian30']]).astype(np.uint8)\n"
        "B = 131; img, noise = S.batch(0, B)\n"
        "eng = P.Engine(256, 256, Bmax=B); eng.synthesize(img, noise, masks, np.arange(B) %% 2); eng.init_state()\n"
        "eng.admm_cnc(6, 0.45, 0.5, 0.05, 64); x = eng.x(); z, w = eng.get_state()\n"
        "eng.init_state(); eng.admm_l1(5, 0.1, 0.015); xl = eng.x()\n"
        "np.savez(sys.argv[1], x=x, z=z, w=w, xl=xl)\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outs = []
    for sched, streams, chunk in (('0', '1', '0'), ('1', '1', '0'), ('0', '2', '0'), ('1', '2', '0'), ('0', '3', '0'), ('0', '1', '48')):
        path = '/tmp/sched_%s_%s_%s.npz' % (sched, streams, chunk)
        subprocess.check_call([sys.executable, '-c', code, path],
                              env=dict(os.environ, PNP_SLICE='0', PNP_FUSED_SCHED=sched, PNP_FUSED_STREAMS=streams, PNP_FUSED_CHUNK=chunk))
        outs.append(np.load(path))
    for o in outs[1:]:
        for k in ('x', 'z', 'w', 'xl'):
            assert np.array_equal(outs[0][k], o[k]), k


def test_device_ssim_matches_reference_definition(P, golden_inputs, golden_admm):
    """pnp_ssim against the oracle's SSIM (valid-region Gaussian 11/1.5, utils_image.py:593-615) and
    the authors' logged values for the committed presets (0.5877 / 0.5600)."""
    gray = golden_inputs['gray']
    mask = golden_inputs['masks']['Q_Random30'].astype(np.float64)
    out, info = P.ADMM_L1(mask, golden_inputs['noises'], images=gray[None], return_info=True, **P.PRESETS['ADMM_L1'])
    assert abs(info['ssim'][0] - O.calculate_ssim(out[0] * 255, gray)) <= 1e-6
    assert abs(info['ssim'][0] - 0.5877) <= 6e-5
    out, info = P.ADMM_CNC(mask, golden_inputs['noises'], images=gray[None], return_info=True, **P.PRESETS['ADMM_CNC'])
    assert abs(info['ssim'][0] - 0.5600) <= 6e-5
    # batched, ragged tile edges (246 = 15*16 + 6) and a 512x512 case
    rng = np.random.default_rng(1)
    for H in (256, 512):
        B = 3
        x = rng.uniform(0, 1, (B, H, H)).astype(np.float32)
        gt = rng.integers(0, 256, (B, H, H), dtype=np.uint8)
        with P.Engine(H, H, Bmax=B) as eng:
            eng.upload(np.zeros((B, H, H), np.complex64), np.ones((H, H), np.uint8))
            import torch
            s = eng.ssim(torch.from_numpy(x).cuda(), gt)
        for b in range(B):
            assert abs(s[b] - O.calculate_ssim(x[b].astype(np.float64) * 255, gt[b])) <= 1e-9


def test_entry_point_reads_testset_and_writes_results_like_the_reference(P, golden_inputs, golden_admm, tmp_path):
    """No `images=`: the solver lists testsets/<Set>, decodes the PNGs to gray, reconstructs all of
    them in one batch, writes results/<Set>_dn_ADMM_CNC/*.png and appends the reference's log lines
    (S4:62-94, 138-172)."""
    import re
    from PIL import Image
    ts = tmp_path / 'testsets' / 'Set1'
    ts.mkdir(parents=True)
    gray = golden_inputs['gray']
    Image.fromarray(gray).save(ts / '05.png')
    Image.fromarray(gray[::-1].copy()).save(ts / '06.png')
    res = tmp_path / 'results'
    mask = golden_inputs['masks']['Q_Random30'].astype(np.float64)
    out = P.ADMM_CNC(mask, golden_inputs['noises'], testsets=str(tmp_path / 'testsets'), results=str(res),
                     **P.PRESETS['ADMM_CNC'])
    assert len(out) == 22 and rel_l2(out[0], golden_admm['cnc_random30_it50']) <= 1e-4
    assert out[1].shape == (256, 256) and out[2].dtype == np.uint8
    d = res / 'Set1_dn_ADMM_CNC'
    assert sorted(p.name for p in d.glob('*.png')) == ['05_ADMM CNC.png', '06_ADMM CNC.png']
    log = (d / 'Set1_dn_ADMM_CNC.log').read_text()
    m = re.search(r'05\.png - PSNR: (\d+\.\d{4}) dB; SSIM: (\d\.\d{4}) ; RE: (\d\.\d{4})\.', log)
    assert m and abs(float(m.group(1)) - 24.5765) <= 0.01 and abs(float(m.group(2)) - 0.5600) <= 1e-4 and abs(float(m.group(3)) - 0.1870) <= 1e-4
    assert re.search(r'------> testset_name: \(Set1\), Average PSNR:\(\d+\.\d{3}\)dB, Average ssim : \(\d\.\d{3}\), Average re : \(\d\.\d{3}\) \)', log)
    saved = np.asarray(Image.open(d / '05_ADMM CNC.png')).astype(np.float64)
    assert np.abs(saved - np.clip(np.rint(out[0] * 255), 0, 255)).max() <= 1


def test_plain_c_consumer_runs_the_loop(P, tmp_path):
    """The gcc-built C program (tests/host/abi_consumer.c) runs synthesis + 5 L1 iterations through
    the C ABI; the Python binding must give the same checksum on the same inputs."""
    import subprocess
    from test_abi_cpu import _build_consumer
    exe = _build_consumer(tmp_path)
    out = subprocess.check_output([exe, 'run']).decode()
    chk = float(out.strip().split()[-1])
    B, H, W = 2, 256, 256
    img = np.zeros((B, H, W), np.float32)
    for b in range(B):
        img[b, 96:160, 64 + 16 * b:192] = 0.5 + 0.25 * b
    mask = np.zeros((H, W), np.uint8)
    rows = [r for r in range(H) if r % 2 == 0 or r < 8 or r > H - 8]
    mask[rows, :] = 1
    with P.Engine(H, W, Bmax=B) as eng:
        eng.synthesize(img, np.zeros((H, W), np.complex64), mask)
        eng.init_state()
        eng.admm_l1(5, 0.1, 0.015)
        x = eng.x()
    assert abs(float(x.astype(np.float64).sum()) - chk) <= 1e-6 * abs(chk)
    assert 'path fused' in out


@pytest.mark.parametrize('seed', range(8))
def test_randomized_configurations_vs_oracle(P, seed):
    """Random batch sizes (odd ones included), random sampling densities and patterns, random
    hyper-parameters and iteration counts, both solvers, fused and generic kernels, 256 and 512."""
    rng = np.random.default_rng(1000 + seed)
    H = 512 if seed % 4 == 3 else 256
    B = int(rng.integers(1, 8))
    K = int(rng.integers(1, 4))
    masks = np.zeros((K, H, H), np.uint8)
    for k in range(K):
        dens = rng.uniform(0.05, 0.9)
        if k % 2 == 0:
            masks[k] = rng.uniform(size=(H, H)) < dens
        else:
            masks[k][rng.uniform(size=H) < dens, :] = 1            # whole rows (Cartesian-like)
        masks[k][0, 0] = 1
    mid = rng.integers(0, K, B).astype(np.int32)
    ys = np.stack([O.synthetic_problem(int(rng.integers(0, 10000)), masks[mid[b]], H, H)[1] for b in range(B)]).astype(np.complex64)
    iters = int(rng.integers(1, 9))
    alpha, lam, reo = rng.uniform(0.2, 1.0), rng.uniform(0.05, 1.0), rng.uniform(0.01, 0.5)
    b_ = float(rng.uniform(0.2, 1.6) / (reo * lam))        # reo*lambda*b <= 1.6 as in the committed presets: beyond that the
                                                            # CNC map amplifies fp32 round-off by > 10x per iteration
    fast = int(seed % 2)
    with P.Engine(H, H, Bmax=B) as eng:
        eng.set_fast_path(fast)
        eng.upload(ys, masks, mid)
        eng.init_state()
        eng.admm_cnc(iters, alpha, lam, reo, b_)
        xc = eng.x()
        eng.init_state()
        eng.admm_l1(iters, lam, reo)
        xl = eng.x()
    for b in range(B):
        y128 = ys[b].astype(np.complex128)
        assert rel_l2(xc[b], O.admm_cnc(y128, masks[mid[b]], iters, alpha, lam, reo, b_)) <= 1e-5, (seed, b)
        assert rel_l2(xl[b], O.admm_l1(y128, masks[mid[b]], iters, lam, reo)) <= 1e-5, (seed, b)


def test_integration_md_ctypes_stub_runs(golden_inputs, golden_admm, tmp_path):
    """The ctypes stub printed in INTEGRATION.md section 2 (what a maintainer of the reference would
    paste) is executed as written -- no package import, no torch -- and must reproduce the
    reference's 50-iteration CNC result."""
    import os
    import re
    import subprocess
    import sys
    from conftest import ROOT
    md = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    code = re.findall(r"```python\n(import ctypes as C.*?)```", md, flags=re.S)[0]
    code = code.replace("os.path.join(os.path.dirname(__file__), 'libpnpmri.so')",
                        repr(os.path.join(ROOT, 'pnp_admm_cnc_mri_amd', 'libpnpmri.so')))
    img = O.requantise(golden_inputs['gray'])[None]
    np.savez(tmp_path / 'in.npz', img=img, mask=golden_inputs['masks']['Q_Random30'], noises=golden_inputs['noises'])
    script = code + (
        "\nimport sys\nd = np.load(sys.argv[1])\n"
        "x = admm_cnc_loop(d['img'], d['mask'], d['noises'], 0.45, 50, 0.5, 0.05, 64)\n"
        "np.save(sys.argv[2], x)\n")
    (tmp_path / 'stub.py').write_text(script)
    subprocess.check_call([sys.executable, str(tmp_path / 'stub.py'), str(tmp_path / 'in.npz'), str(tmp_path / 'x.npy')])
    x = np.load(tmp_path / 'x.npy')
    assert rel_l2(x[0], golden_admm['cnc_random30_it50']) <= 1e-4
