"""CPU checks of the denoiser declarations and the host logic around them (no GPU needed)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLD
from pnp_admm_cnc_mri_amd import denoisers as D


def test_state_dict_keys_match_reference_models():
    """Key names and shapes recorded when the reference's own model classes loaded these weights
    with strict=True (oracle/make_golden_pnp.py)."""
    shapes = json.load(open(os.path.join(GOLD, 'pnp_known.json')))['state_dict_shapes']
    for name, ref in shapes.items():
        net, _, _ = D.build(name)
        mine = {k: list(v.shape) for k, v in net.state_dict().items()}
        assert mine == ref, name
    n_params = lambda n: sum(v.numel() for v in D.build(n)[0].state_dict().values())
    assert n_params('ffdnet_gray') == 485316 and n_params('drunet_gray') == 32638656 and n_params('dncnn_15') == 555137


def test_seeded_weights_are_deterministic_and_torch_rng_free():
    net, _, _ = D.build('ffdnet_gray')
    torch.manual_seed(1)
    a = D.seeded_state_dict(net, 7)
    torch.manual_seed(2)
    b = D.seeded_state_dict(D.build('ffdnet_gray')[0], 7)
    assert all(torch.equal(a[k], b[k]) for k in a)
    c = D.seeded_state_dict(net, 8)
    assert not torch.equal(a['model.0.weight'], c['model.0.weight'])


def test_family_dispatch_and_constants():
    assert D.family('dncnn_15') == 'dncnn' and D.family('fdncnn_gray') == 'fdncnn'
    assert D.family('drunet_gray') == 'drunet' and D.family('ircnn_gray') == 'ircnn' and D.family('ffdnet_gray_clip') == 'ffdnet'
    assert D.build('dncnn_15')[1] == 15 and D.build('drunet_gray')[1] == 15 / 255.0 and D.build('drunet_gray')[2]
    assert len([m for m in D.build('dncnn_gray_blind')[0].model if isinstance(m, torch.nn.Conv2d)]) == 20


def test_ffdnet_batched_sigma_and_odd_sizes():
    net, _, _ = D.build('ffdnet_gray')
    net.load_state_dict(D.seeded_state_dict(net, 1))
    net.eval()
    x = torch.rand(3, 1, 31, 34)
    with torch.no_grad():
        y = net(x, torch.full((1, 1, 1, 1), 15 / 255.))
        y1 = torch.cat([net(x[i:i + 1], torch.full((1, 1, 1, 1), 15 / 255.)) for i in range(3)])
    assert y.shape == x.shape and torch.allclose(y, y1, atol=1e-6)


def test_augment_modes_invert_like_the_reference():
    x = torch.arange(2 * 1 * 4 * 4, dtype=torch.float32).reshape(2, 1, 4, 4)
    for i in range(8):
        back = D.augment_img_tensor4(D.augment_img_tensor4(x, i), 8 - i if i in (3, 5) else i)
        assert torch.equal(back, x), i


def test_split_fn_quadrants_cover_image():
    """512x512 goes through four overlapping 288x288 quadrants (utils/utils_model.py:91-108);
    with a 1x1 'network' the stitched result must equal the whole-image result."""
    model = lambda t: 2.0 * t[:, :1] + 1.0
    L = torch.rand(2, 2, 512, 512)
    E = D.test_split_fn(model, L, refield=32, min_size=256, modulo=16)
    assert torch.allclose(E, 2.0 * L[:, :1] + 1.0)
    L = torch.rand(1, 2, 250, 256)
    assert D.test_split_fn(model, L, refield=32, min_size=256, modulo=16).shape == (1, 1, 250, 256)


def test_ircnn_bank_index():
    sig = torch.tensor([49.0, 30.1, 15.0]) / 255.
    den = D.Denoiser('ircnn_gray', D.build('ircnn_gray')[0], 15 / 255.0, sigmas=sig, bank=None)
    assert [int(np.ceil(float(s) * 255. / 2.) - 1) for s in sig] == [24, 15, 7]
    den.select_bank(0)       # no bank: no-op


def _tiny_net(seed=5):
    """the 3-layer conv net of oracle/make_golden_pnp.py::tiny_net (same NumPy-seeded weights)"""
    rng = np.random.default_rng(seed)
    net = torch.nn.Sequential(torch.nn.Conv2d(2, 8, 3, 1, 1), torch.nn.ReLU(), torch.nn.Conv2d(8, 8, 3, 1, 1),
                              torch.nn.ReLU(), torch.nn.Conv2d(8, 1, 3, 1, 1))
    with torch.no_grad():
        for p_ in net.parameters():
            p_.copy_(torch.from_numpy(rng.standard_normal(tuple(p_.shape)).astype(np.float32) * 0.3))
    return net.eval()


@pytest.fixture(scope='module')
def pnp_golden():
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    import json
    return (np.load(os.path.join(here, 'golden', 'pnp_set1_05.npz')),
            json.load(open(os.path.join(here, 'golden', 'pnp_known.json')))['known'])


@pytest.mark.parametrize('tag', ['split_1level', 'split_2level', 'split_whole_padded'])
def test_split_fn_matches_the_reference(pnp_golden, tag):
    """The batched quadrant forward against outputs of the reference's utils_model.test_mode(mode=2)
    (utils/utils_model.py:76-109) on the same seeded conv net, whose zero padding makes every output
    pixel depend on which window produced it: one-level split, recursive split, padded whole image."""
    arrays, known = pnp_golden
    net = _tiny_net()
    calls = []
    model = lambda t: (calls.append(tuple(t.shape)), net(t))[1]
    with torch.no_grad():
        got = D.test_mode(model, torch.from_numpy(arrays[tag + '_in']), mode=2, sf=1, **known[tag + '_kw'])
    want = arrays[tag + '_out']
    assert got.shape == want.shape
    # batched vs per-window conv calls differ in summation order (~2e-6 on O(1) values); a misplaced window is O(1)
    assert np.abs(got.numpy() - want).max() <= 1e-5, np.abs(got.numpy() - want).max()
    B = arrays[tag + '_in'].shape[0]
    if tag == 'split_1level':
        assert calls == [(4 * B, 2, 24, 32)]              # the four windows in ONE model call
    elif tag == 'split_2level':
        assert calls == [(64 * B, 2, 24, 24)]             # 80x96 -> 48x56 -> 32x32 -> 24x24 windows: 4^3 of them, still one call
    else:
        assert calls == [(B, 2, 32, 28)]                  # whole image, replicate-padded to modulo 4


def test_augment_modes_match_the_reference(pnp_golden):
    """all eight x8 views against the reference's augment_img_tensor4 (utils/utils_image.py:333-349)
    on a non-square tensor (shape changes for the odd quarter turns)."""
    arrays, _ = pnp_golden
    a = torch.arange(2 * 1 * 5 * 7, dtype=torch.float32).reshape(2, 1, 5, 7)
    for m in range(8):
        assert np.array_equal(D.augment_img_tensor4(a, m).numpy(), arrays['augment_mode%d' % m]), m


def test_denoiser_call_restores_the_cudnn_benchmark_flag():
    """Denoiser.__call__ may switch MIOpen's find mode on around its forward passes (miopen_find); the process-wide flag must
    come back as it was, whatever the setting and also when the forward raises."""
    import torch
    from pnp_admm_cnc_mri_amd import denoisers as D
    net, nlm, _ = D.build('ffdnet_gray')
    net.load_state_dict(D.seeded_state_dict(net, 3))
    x = torch.rand(2, 1, 32, 32)
    for before in (False, True):
        torch.backends.cudnn.benchmark = before
        for mode in ('auto', True, False):
            den = D.Denoiser('ffdnet_gray', net.eval(), nlm, miopen_find=mode)
            y = den(x, 0)
            assert y.shape == x.shape and torch.backends.cudnn.benchmark is before

        class Boom(D.Denoiser):
            def _one(self, x, i):
                raise RuntimeError('boom')
        with pytest.raises(RuntimeError):
            Boom('ffdnet_gray', net.eval(), nlm, miopen_find=True)(x, 0)
        assert torch.backends.cudnn.benchmark is before
    torch.backends.cudnn.benchmark = False


@pytest.mark.parametrize('name,gflop', [('ffdnet_gray', 15.9), ('dncnn_15', 72.6), ('drunet_gray', 277.0)])
def test_forward_flops_match_the_survey(name, gflop):
    """bench_pnp.py prices the denoiser at 2 x the MACs its convolutions really execute (hooks on a one-slice call);
    the figures of SURVEY.md section 8 (a12): FFDNet 15.9, DnCNN-17 72.6, DRUNet 277 GFLOP per 256 x 256 forward."""
    import torch
    from pnp_admm_cnc_mri_amd import denoisers as D
    net, nlm, sched = D.build(name)
    sig = torch.full((4,), 0.1) if sched else None
    den = D.Denoiser(name, net.eval(), nlm, sigmas=sig, noises=np.zeros((256, 256), np.complex64), miopen_find=False)
    got = D.forward_flops(den, 256, 256, 'cpu') / 1e9
    assert abs(got - gflop) <= 0.02 * gflop, got


def test_hip_backend_needs_gpu_tensors_and_float32():
    """`Denoiser(backend='hip')` never computes on the CPU: a CPU tensor raises; autocast modes are refused; the default
    backend is PyTorch."""
    import pytest as _pytest
    from pnp_admm_cnc_mri_amd import denoisers as D
    net, nlm, _ = D.build('ffdnet_gray')
    net.load_state_dict(D.seeded_state_dict(net, 1))
    assert D.Denoiser('ffdnet_gray', net.eval(), nlm).backend == 'torch' and net.backend == 'torch'
    den = D.Denoiser('ffdnet_gray', net.eval(), nlm, backend='hip', miopen_find=False)
    assert net.backend == 'hip'
    with _pytest.raises(RuntimeError, match='CUDA'):
        den(torch.rand(1, 1, 32, 32), 0)
    with _pytest.raises(ValueError):
        D.Denoiser('ffdnet_gray', net, nlm, backend='hip', cnn_dtype='bf16')
    with _pytest.raises(ValueError):
        D.Denoiser('ffdnet_gray', net, nlm, backend='cuda')
    # the body layers the kernel takes: 13 of FFDNet's 15, 15 of DnCNN-17's, IRCNN's five dilated ones
    count = lambda m: sum(1 for c in m.model if D._hip_body_ok(c))
    assert count(net) == 13 and count(D.build('dncnn_15')[0]) == 15 and count(D.build('ircnn_gray')[0]) == 5
    assert all(D.hip_covers_stack(D.build(n)[0].model) for n in ('ffdnet_gray', 'dncnn_15', 'fdncnn_gray', 'ircnn_gray'))


def test_f16x3_backend_selection_on_the_cpu():
    """`backend='hip_f16x3'`: same rules as 'hip' (no CPU computation, float32 only); which layers it takes -- the 64-channel
    ones of every family as 'hip' does, and DRUNet's 128 / 256 / 512-channel residual blocks on top (dilation 1 only)"""
    import pytest as _pytest
    import torch.nn as nn
    from pnp_admm_cnc_mri_amd import denoisers as D
    net, nlm, _ = D.build('dncnn_15')
    den = D.Denoiser('dncnn_15', net.eval(), nlm, backend='hip_f16x3', miopen_find=False)
    assert net.backend == 'hip_f16x3' and D._hip_math(net.backend) == 'f16x3' and D._hip_math('hip') == 'f32'
    with _pytest.raises(RuntimeError, match='CUDA'):
        den(torch.rand(1, 1, 32, 32), 0)
    with _pytest.raises(ValueError):
        D.Denoiser('dncnn_15', net, nlm, backend='hip_f16x3', cnn_dtype='fp16')
    unet, nlm2, _ = D.build('drunet_gray')
    D.Denoiser('drunet_gray', unet.eval(), nlm2, sigmas=torch.tensor([0.1]), backend='hip_f16x3')
    blocks = [m for m in unet.modules() if isinstance(m, D._ResBlock)]
    assert len(blocks) == 28 and all(m.backend == 'hip_f16x3' for m in blocks)
    taken = lambda math: sum(1 for m in blocks if D._hip_body_ok(m.res[0], math) and D._hip_body_ok(m.res[2], math))
    assert taken('f32') == 8 and taken('f16x3') == 28            # float32 kernel: the 64-channel scale only
    assert not D._hip_body_ok(nn.Conv2d(128, 128, 3, 1, 2, dilation=2), 'f16x3')      # wide layers: dilation 1 only
    assert not D._hip_body_ok(nn.Conv2d(96, 96, 3, 1, 1), 'f16x3') and not D._hip_body_ok(nn.Conv2d(128, 64, 3, 1, 1), 'f16x3')
    assert D._hip_body_ok(nn.Conv2d(1024, 1024, 3, 1, 1), 'f16x3') and not D._hip_body_ok(nn.Conv2d(1088, 1088, 3, 1, 1), 'f16x3')


# ----------------------------------------------------------------------------------------------
# The oracle's PnP loops against the UNMODIFIED reference at the presets' own 50 iterations (tests/golden/pnp50_set1_05.npz,
# contractive fixture weights): both sides run the CNN in CPU PyTorch and the x-update in NumPy, so the oracle -- the checker of
# every GPU PnP test and of bench_pnp.py's parity key -- must land on the reference's x to float32 round-off, IRCNN's bank switch
# (S6:289-298 under the np.int shim) included.  The cheap families only: the CPU suite stays within minutes.
# ----------------------------------------------------------------------------------------------
def _cpu_denoiser(name, iters, noises):
    from conftest import weights50
    from pnp_admm_cnc_mri_amd import utils_pnp
    net, nlm, sched = D.build(name)
    w = weights50(name)
    bank = w if D.family(name) == 'ircnn' else None
    net.load_state_dict(bank['0'] if bank else w)
    sig = torch.tensor(utils_pnp.get_rho_sigma(max(0.255 / 255., nlm), iters, 49, nlm * 255., 1.0)[1]) if sched else None
    den = D.Denoiser(name, net.eval(), nlm, sigmas=sig, noises=noises, bank=bank, x8=False)

    def denoise(a, i):
        den.select_bank(i)
        with torch.no_grad():
            return den._one(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))[None, None], i)[0, 0].numpy()
    return den, denoise


@pytest.mark.parametrize('name,loop', [('ffdnet_gray', 'cnc'), ('ircnn_gray', 'cnc'), ('ircnn_gray', 'l1')])
def test_oracle_pnp_loops_at_fifty_iterations_vs_the_unmodified_reference(golden_inputs, name, loop):
    from conftest import rel_l2
    from oracle import admm_oracle as O
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    meta = json.load(open(os.path.join(GOLD, 'pnp_known.json')))['known50']
    gold = np.load(os.path.join(GOLD, 'pnp50_set1_05.npz'))
    mask = golden_inputs['masks']['Q_Random30'].astype(np.float64)
    y = O.synthesize(np.float32(golden_inputs['gray'] / 255.), mask, golden_inputs['noises'])
    den, denoise = _cpu_denoiser(name, 50, golden_inputs['noises'])
    if loop == 'cnc':
        o = meta['cnc_d_%s_opts' % name]
        assert int(o['iter_num']) == 50
        x = O.pnp_admm_cnc(y, mask, denoise, 50, o['alpha'], o['lambda1'], o['reo'], o['b'])
        ref, line = gold['cnc_d_' + name], meta['cnc_d_' + name]
    else:
        o = meta['l1_d_%s_opts' % name]
        x = O.pnp_admm_l1(y, mask, denoise, 50, o['reo'])
        ref, line = gold['l1_d_' + name], meta['l1_d_' + name]
    assert rel_l2(x, ref) <= 2e-6, rel_l2(x, ref)
    if name == 'ircnn_gray':
        assert den.former_idx == 7                                   # the bank was walked down from model 24 to model 7 (sigma 49 -> 15)
    psnr = O.calculate_psnr(np.round(x.astype(np.float64) * 255.), golden_inputs['gray'])
    assert ('PSNR: %.4f dB' % psnr if loop == 'cnc' else 'PSNR: %.2f dB' % psnr) in line, (psnr, line)


def test_trained_fixture_network_loads_strictly_and_denoises():
    """tests/golden/ffdnet_gray_trained.npz (oracle/train_fixture_denoiser.py): KAIR's FFDNet keys and shapes, finite weights well inside
    the half range, and a network that does what its training record says -- on a seeded image with sigma = 25 noise it gains several dB."""
    from conftest import weights_trained
    from pnp_admm_cnc_mri_amd import synthetic as S
    sd = weights_trained()
    net, _, _ = D.build('ffdnet_gray')
    net.load_state_dict(sd, strict=True)
    net.eval()
    assert all(bool(torch.isfinite(v).all()) and float(v.abs().max()) < 10 for v in sd.values())
    rec = json.load(open(os.path.join(GOLD, 'pnp_known.json')))['trained']
    assert rec['training']['held_out_psnr']['25']['denoised'] - rec['training']['held_out_psnr']['25']['noisy'] > 8 and rec['lipschitz_at_x0'] > 1
    clean = torch.from_numpy(S.phantom(7))[None, None]
    noisy = clean + (25 / 255.) * torch.from_numpy(np.random.default_rng(1).standard_normal(clean.shape).astype(np.float32))
    with torch.no_grad():
        den = net(noisy, torch.full((1, 1, 1, 1), 25 / 255.))
    mse = lambda a: float(((a - clean) ** 2).mean())
    assert 10 * np.log10(mse(noisy) / mse(den)) > 6


@pytest.mark.parametrize('n_it', [5, 20])
def test_oracle_pnp_loop_with_the_trained_network_vs_the_unmodified_reference(golden_inputs, n_it):
    """the oracle's PNP_ADMM_CNC_D loop driven by the TRAINED FFDNet on the CPU against the unmodified S6's output with the same weights
    (tests/golden/pnp50_set1_05.npz: trained_cnc_d_ffdnet_gray_it<n>): same CPU convolutions, same NumPy transforms -> float32 round-off"""
    from conftest import rel_l2, weights_trained
    from oracle import admm_oracle as O
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    meta = json.load(open(os.path.join(GOLD, 'pnp_known.json')))['known50']
    gold = np.load(os.path.join(GOLD, 'pnp50_set1_05.npz'))
    mask = golden_inputs['masks']['Q_Random30'].astype(np.float64)
    y = O.synthesize(np.float32(golden_inputs['gray'] / 255.), mask, golden_inputs['noises'])
    net, nlm, _ = D.build('ffdnet_gray')
    net.load_state_dict(weights_trained())
    den = D.Denoiser('ffdnet_gray', net.eval(), nlm)

    def denoise(a, i):
        with torch.no_grad():
            return den._one(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))[None, None], i)[0, 0].numpy()
    o = meta['trained_cnc_d_ffdnet_gray_it%d_opts' % n_it]
    x = O.pnp_admm_cnc(y, mask, denoise, n_it, o['alpha'], o['lambda1'], o['reo'], o['b'])
    assert rel_l2(x, gold['trained_cnc_d_ffdnet_gray_it%d' % n_it]) <= 2e-6, rel_l2(x, gold['trained_cnc_d_ffdnet_gray_it%d' % n_it])


def test_auto_backend_without_a_hip_device_is_torch():
    """cnn_backend='auto' (the entry points' default): no HIP device here -> the PyTorch backend, with the reason; autocast modes likewise"""
    net, _, _ = D.build('ffdnet_gray')
    be, why = D.auto_backend(net)
    assert be == 'torch' and 'HIP' in why
    assert D.auto_backend(net, cnn_dtype='bf16')[0] == 'torch'


def test_trained_dncnn_fixture_and_the_oracle_pair_loop_vs_the_unmodified_reference(golden_inputs):
    """tests/golden/dncnn_25_trained.npz: KAIR's DnCNN-17 keys and shapes, a network that really denoises (the committed training record);
    the oracle's DnCNN-pair loop (S6:485-525) driven by it on the CPU against the unmodified S6's output with the same weights at 5 iterations."""
    from conftest import rel_l2, weights_trained
    from oracle import admm_oracle as O
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    net, nlm, _ = D.build('dncnn_25')
    net.load_state_dict(weights_trained('dncnn_25'), strict=True)
    meta = json.load(open(os.path.join(GOLD, 'pnp_known.json')))
    tr = meta['trained_dncnn']['training']['held_out_psnr']['25']
    assert tr['denoised'] - tr['noisy'] >= 10.0
    gold = np.load(os.path.join(GOLD, 'pnp50_set1_05.npz'))
    mask = golden_inputs['masks']['Q_Random30'].astype(np.float64)
    y = O.synthesize(np.float32(golden_inputs['gray'] / 255.), mask, golden_inputs['noises'])
    den = D.Denoiser('dncnn_25', net.eval(), nlm)

    def denoise(a, i):
        with torch.no_grad():
            return den._one(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))[None, None], i)[0, 0].numpy()
    o = meta['known50']['trained_cnc_dncnn_pair_it5_opts']
    x = O.pnp_admm_cnc(y, mask, denoise, 5, o['alpha'], o['lambda1'], o['reo'], o['b'], denoise2=denoise)
    assert rel_l2(x, gold['trained_cnc_dncnn_pair_it5']) <= 2e-6, rel_l2(x, gold['trained_cnc_dncnn_pair_it5'])
