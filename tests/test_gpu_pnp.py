"""GPU parity of the PnP entry points (PNP_ADMM_CNC_D, PNP_ADMM_CNC_DnCNN, PNP_ADMM_L1_D).

Two layers of evidence:
  (1) against golden x produced by the UNMODIFIED reference scripts on CPU with the same seeded
      weights (oracle/make_golden_pnp.py; 05.png, Q_Random30, 3 iterations).  The reference runs
      the CNN on CPU-PyTorch, here it runs on MIOpen: conv rounding differs at 1e-6 per forward and
      random-weight nets are not contractive, so the tolerance is 2e-4 relative L2 (PSNR 0.01 dB);
  (2) against the oracle's PnP loop (oracle.pnp_admm_cnc / pnp_admm_l1: float64 NumPy x-update,
      reference marshalling semantics) driven with the SAME GPU denoiser as callback, which
      isolates the HIP x-update + glue kernels: relative L2 <= 1e-5.
"""
import json
import os

import numpy as np
import pytest

from oracle import admm_oracle as O
from conftest import rel_l2, GOLD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    import pnp_admm_cnc_mri_amd as P
    from pnp_admm_cnc_mri_amd import solvers_pnp, denoisers, _lib
    assert _lib.device_count() >= 1 and torch.cuda.is_available()
    torch.backends.cudnn.benchmark = False
    torch.backends.cudnn.deterministic = True
    known = json.load(open(os.path.join(GOLD, 'pnp_known.json')))
    gold = np.load(os.path.join(GOLD, 'pnp_set1_05.npz'))
    return dict(torch=torch, P=P, S=solvers_pnp, D=denoisers, known=known['known'], gold=gold)


def _weights(env, name):
    net, _, _ = env['D'].build(name)
    return env['D'].seeded_state_dict(net, env['known']['seeds'][name])


def _psnr_close(x, ref, gt):
    return abs(O.calculate_psnr(np.round(x.astype(np.float64) * 255), gt) - O.calculate_psnr(np.round(ref.astype(np.float64) * 255), gt)) <= 0.01


@pytest.mark.parametrize('backend', ['torch', 'hip', 'hip_f16x3'])
@pytest.mark.parametrize('name', ['ffdnet_gray', 'fdncnn_gray', 'drunet_gray'])
def test_pnp_admm_cnc_d_golden(env, golden_inputs, name, backend, tmp_path):
    """against the UNMODIFIED reference script's output, with the CNN forward on every backend: PyTorch / MIOpen, the float32-MFMA
    kernels, the split-half f16 kernels -- the same bar for all three"""
    opts = dict(env['known']['cnc_d_%s_opts' % name])
    opts['iter_num'] = int(opts['iter_num'])
    mask = golden_inputs['masks']['Q_Random30'].astype(np.float64)
    out, psnr1 = env['S'].PNP_ADMM_CNC_D(name, mask, golden_inputs['noises'], images=golden_inputs['gray'][None],
                                         model=_weights(env, name), results=str(tmp_path), cnn_backend=backend, **opts)
    ref = env['gold']['cnc_d_' + name]
    assert len(out) == 22 and out[0].shape == (256, 256)
    assert rel_l2(out[0], ref) <= 2e-4, rel_l2(out[0], ref)
    assert _psnr_close(out[0], ref, golden_inputs['gray'])
    # device PSNR of the uint8-quantised image (S6:314); float32 vs float64 rounding of x*255 may flip a pixel at .5
    assert abs(psnr1[0] - O.calculate_psnr(np.round(out[0] * 255), golden_inputs['gray'])) <= 1e-4


def test_pnp_admm_cnc_dncnn_pair_golden(env, golden_inputs, tmp_path):
    opts = dict(env['known']['cnc_dncnn_pair_opts'])
    opts['iter_num'] = int(opts['iter_num'])
    mask = golden_inputs['masks']['Q_Random30'].astype(np.float64)
    # the reference loads dncnn_25's file into BOTH nets (S6:435)
    out, _ = env['S'].PNP_ADMM_CNC_DnCNN('dncnn_25', 'dncnn_15', mask, golden_inputs['noises'],
                                         images=golden_inputs['gray'][None], model=_weights(env, 'dncnn_25'),
                                         results=str(tmp_path), **opts)
    ref = env['gold']['cnc_dncnn_pair']
    assert rel_l2(out[0], ref) <= 2e-4, rel_l2(out[0], ref)


@pytest.mark.parametrize('name', ['ffdnet_gray', 'dncnn_15', 'fdncnn_gray', 'drunet_gray'])
def test_pnp_admm_l1_d_golden(env, golden_inputs, name, tmp_path):
    opts = dict(env['known']['l1_d_%s_opts' % name])
    opts['iter_num'] = int(opts['iter_num'])
    mask = golden_inputs['masks']['Q_Random30'].astype(np.float64)
    out = env['S'].PNP_ADMM_L1_D(name, mask, golden_inputs['noises'], images=golden_inputs['gray'][None],
                                 model=_weights(env, name), results=str(tmp_path), **opts)
    ref = env['gold']['l1_d_' + name]
    assert rel_l2(out[0], ref) <= 2e-4, rel_l2(out[0], ref)


@pytest.mark.parametrize('name', ['ffdnet_gray', 'dncnn_15'])
def test_pnp_batched_vs_oracle_loop(env, golden_inputs, name, tmp_path):
    """B = 3 slices with three different masks: the batched device loop equals the oracle's
    per-slice loop when both call the same GPU denoiser."""
    torch, D = env['torch'], env['D']
    masks = np.stack([golden_inputs['masks'][k] for k in ('Q_Random30', 'Q_Radial30', 'Q_Cartesian30')]).astype(np.uint8)
    B = 3
    mid = np.arange(B, dtype=np.int32)
    imgs, ys = [], []
    for b in range(B):
        img, y = O.synthetic_problem(b, masks[mid[b]])
        imgs.append(img)
        ys.append(y.astype(np.complex64))
    ys = np.stack(ys)
    sd = _weights(env, name)
    net, nlm, _ = D.build(name)
    net.load_state_dict(sd)
    den = D.Denoiser(name, net.eval(), nlm).to(torch.device('cuda'))

    def denoise(a, i):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))[None, None].cuda()
        return den(t, i)[0, 0].cpu().numpy()

    alpha, lam, reo, b_ = 0.9, 1.35, 0.45, 0.3
    out, _ = env['S'].PNP_ADMM_CNC_D(name, masks, None, y=ys, mask_id=mid, model=sd, results=str(tmp_path),
                                     alpha=alpha, iter_num=4, lambda1=lam, reo=reo, b=b_)
    outl = env['S'].PNP_ADMM_L1_D(name, masks, None, y=ys, mask_id=mid, model=sd, results=str(tmp_path), iter_num=4, reo=0.25)
    for b in range(B):
        y128 = ys[b].astype(np.complex128)
        ref = O.pnp_admm_cnc(y128, masks[mid[b]], denoise, 4, alpha, lam, reo, b_)
        assert rel_l2(out[b], ref) <= 1e-5, (b, rel_l2(out[b], ref))
        refl = O.pnp_admm_l1(y128, masks[mid[b]], denoise, 4, 0.25)
        assert rel_l2(outl[b], refl) <= 1e-5, (b, rel_l2(outl[b], refl))


def test_glue_kernels(env):
    """pnp_cnc_combine / pnp_dual_clamp / pnp_add against the torch expressions of S6:301-308."""
    torch, P = env['torch'], env['P']
    g = torch.Generator(device='cuda').manual_seed(3)
    B = 2
    z, x, w, s = (torch.rand((B, 1, 256, 256), device='cuda', generator=g) * 1.4 - 0.2 for _ in range(4))
    alpha, lam, reo, b = 0.9, 1.35, 0.45, 0.3
    with P.Engine(256, 256, Bmax=B) as eng:
        eng.upload(np.zeros((B, 256, 256), np.complex64), np.ones((256, 256), np.uint8))
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        t = torch.empty_like(z)
        eng.cnc_combine(z, x, w, s, t, alpha, lam, reo, b)
        ref = (1 - alpha) * z + alpha * (x + w) + alpha * reo * lam * b * (z - s)
        assert torch.allclose(t, ref, rtol=0, atol=2e-7)
        a = torch.empty_like(z)
        eng.add(x, w, a)
        assert torch.equal(a, x + w)
        x2, z2, w2 = x.clone(), z.clone(), w.clone()
        eng.dual_clamp(x2, z2, w2)
        wr = (w + x - z)
        assert torch.equal(x2, x.clamp(0, 1)) and torch.equal(z2, z.clamp(0, 1)) and torch.equal(w2, wr.clamp(0, 1))


def test_config5_shape_512_mixed_masks_drunet(env, tmp_path):
    """config 5's shape: PNP_ADMM_CNC_D at 512x512 with a bank of three masks (mask_id = b % 3) and
    DRUNet going through the four-quadrant split of test_mode(mode=2) (utils/utils_model.py:91-108).
    Checked against the oracle loop driven with the same GPU denoiser."""
    torch, D = env['torch'], env['D']
    H = W = 512
    masks = np.stack([O.synthetic_mask(k, H, W) for k in ('random', 'radial', 'cartesian')])
    B = 3
    mid = np.arange(B, dtype=np.int32) % 3
    ys = np.stack([O.synthetic_problem(b, masks[mid[b]], H, W)[1] for b in range(B)]).astype(np.complex64)
    name = 'drunet_gray'
    net, nlm, _ = D.build(name)
    sd = D.seeded_state_dict(net, 5)
    net.load_state_dict(sd)
    iters = 2
    from pnp_admm_cnc_mri_amd import utils_pnp
    sig = torch.tensor(utils_pnp.get_rho_sigma(max(0.255 / 255., nlm), iters, 49, nlm * 255., 1.0)[1])
    den = D.Denoiser(name, net.eval(), nlm, sigmas=sig).to(torch.device('cuda'))

    def denoise(a, i):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))[None, None].cuda()
        return den(t, i)[0, 0].cpu().numpy()

    opts = dict(alpha=1, iter_num=iters, lambda1=0.8, reo=0.8, b=0.45)           # S6:577 preset, 2 iterations
    out, _ = env['S'].PNP_ADMM_CNC_D(name, masks, None, y=ys, mask_id=mid, model=sd, results=str(tmp_path), **opts)
    # the same with one slice per CNN call: the convolutions then have the oracle loop's shapes (MIOpen picks its kernels, and
    # with them the summation order, by shape), and the north-star bar holds; batched calls differ from it by float32 round-off
    # that two passes through a random-weight U-Net amplify to ~1.5e-5
    out1, _ = env['S'].PNP_ADMM_CNC_D(name, masks, None, y=ys, mask_id=mid, model=sd, results=str(tmp_path), cnn_batch=1, **opts)
    for b in range(B):
        ref = O.pnp_admm_cnc(ys[b].astype(np.complex128), masks[mid[b]], denoise, iters, 1, 0.8, 0.8, 0.45)
        assert out[b].shape == (H, W)
        assert rel_l2(out1[b], ref) <= 1e-5, (b, rel_l2(out1[b], ref))
        assert rel_l2(out[b], ref) <= 2e-5, (b, rel_l2(out[b], ref))
