"""GPU tests added in round 2: the self-launching multi-GPU bench, the reference's file names / log
formats for every solver, the .mat input path, zero-iteration calls, unsampled NaNs, device-side
argument validation, and the PnP coverage holes (IRCNN bank, all eight x8 modes, full-batch config 3)."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from oracle import admm_oracle as O
from conftest import rel_l2, GOLD, ROOT

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


# ----------------------------------------------------------------------------------------------
# bench.py launches its own ranks
# ----------------------------------------------------------------------------------------------
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun around it: the parent starts two rank processes as
    children (both on this box's one GPU, gloo rendezvous: --rehearse-gloo), relays rank 0's single
    JSON line and exits 0.  The N = 1 invocation keeps its shape."""
    env_ = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--rehearse-gloo', '--batch', '32', '--steps', '3',
           '--warmup', '1', '--sustain-s', '0.2']
    r = subprocess.run(cmd, env=env_, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['steps'] == 3 and j['scaling'] == 'weak' and j['x_finite']
    assert j['gather_ms'] is not None and j['cpu_baseline'] is None
    # every rank's own clocks are in the line (a slow GPU of an 8-GPU job must be visible); the job's time is their MAX
    assert len(j['per_rank']['ms_per_step']) == 2 and abs(max(j['per_rank']['ms_per_step']) - j['ms_per_step']) <= 1e-9
    assert len(j['per_rank']['sustained_ms_per_step']) == 2 and j['sustained']['value'] > 0
    assert abs(max(j['per_rank']['sustained_ms_per_step']) - j['sustained']['ms_per_step']) <= 1e-9
    assert j['value'] > 0 and abs(j['value'] - 2 * 3 / (j['ms_per_step'] * 3e-3) * 32 / 512) <= 1e-6 * j['value']
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--batch', '32', '--steps', '3', '--warmup', '1',
                         '--no-cpu-baseline', '--sustain-s', '0'], env=env_, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r1.returncode == 0, r1.stderr.decode()[-2000:]
    j1 = json.loads([l for l in r1.stdout.decode().splitlines() if l.startswith('{')][0])
    assert j1['n_gpus'] == 1 and j1['gather_ms'] is None and set(j1['roofline']) >= {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'}
    assert j1['sustained'] is None and j1['per_rank'] is None
    # same slices on rank 0 of both runs: the N = 2 job's rank 0 did exactly the N = 1 job's work
    assert j1['x_checksum'] == j['x_checksum']


# ----------------------------------------------------------------------------------------------
# file-based outputs: names and log formats of all solvers
# ----------------------------------------------------------------------------------------------
def _testset(tmp_path, gray):
    from PIL import Image
    ts = tmp_path / 'testsets' / 'Set1'
    ts.mkdir(parents=True)
    Image.fromarray(gray).save(ts / '05.png')
    return str(tmp_path / 'testsets'), tmp_path / 'results'


def test_bench_pnp_sub_record():
    """bench.py's `pnp` sub-record (BASELINE.json configs[2] on the PyTorch / MIOpen backend and on the f16x3 HIP backend, two child
    runs of bench_pnp.py): both children report a finite result, the HIP backend's is the faster one, a failing child would be
    reported inside the record instead of failing the line."""
    sys.path.insert(0, ROOT)
    import bench
    os.environ['PNP_BENCH_PNP_SUSTAIN_S'] = '1'                       # the line's 5 s per f16x3 child would be 8 s of this suite
    try:
        rec = bench.pnp_record(steps=1, warmup=1)
    finally:
        del os.environ['PNP_BENCH_PNP_SUSTAIN_S']
    assert set(rec) >= {'config', 'unit', 'torch', 'hip_f16x3', 'config4_shard_drunet_hip_f16x3', 'speedup', 'note'}, rec
    # round 6: the f16x3 children carry a `sustained` twin (back-to-back iterations, no host sync inside a chunk); MIOpen's does not
    for b in ('hip_f16x3', 'config4_shard_drunet_hip_f16x3'):
        su = rec[b]['sustained']
        assert su['span_s'] >= 1.0 and su['value'] > 0 and abs(su['value'] - su['steps'] / su['span_s']) <= 1e-9 * su['value'], su
    assert rec['torch']['sustained'] is None
    assert 'drunet_gray' in rec['config4_shard_drunet_hip_f16x3']['workload'] and 'Q_Cartesian30' in rec['config4_shard_drunet_hip_f16x3']['workload']
    for b in ('torch', 'hip_f16x3', 'config4_shard_drunet_hip_f16x3'):
        assert 'error' not in rec[b] and rec[b]['x_finite'] and rec[b]['denoiser_outputs_finite'] and rec[b]['value'] > 0, rec[b]
        # every child re-proves parity of what it timed: three slices against the oracle's loop with the same denoiser
        assert rec[b]['parity']['slices'] == [0, 255, 511] and max(rec[b]['parity']['rel_l2_vs_oracle']) <= 1e-5, rec[b]['parity']
        rf = rec[b]['denoiser_roofline']
        assert 0 < rf['frac'] <= 1 and abs(rf['frac'] - rf['achieved'] / rf['peak']) <= 1e-12, rf      # a PHYSICAL fraction
    assert rec['torch']['denoiser_roofline']['bound'] == 'mfma_f32' and rec['hip_f16x3']['denoiser_roofline']['bound'] == 'mfma_f16'
    assert rec['hip_f16x3']['denoiser_roofline']['peak'] == 2.5e6 and rec['hip_f16x3']['denoiser_roofline']['frac_fp32_equivalent'] > 1.0
    assert rec['speedup'] > 1.5, rec


def test_file_names_and_log_formats_follow_each_reference_script(env, golden_inputs, tmp_path):
    """S1:138/150 ('_PDG L1', PSNR with 2 decimals), S4:144/155 ('_ADMM CNC', 4 decimals),
    S3:308/320 ('_<model>_PNP_ADMM_L1_D', 2 decimals), S6:320/332 ('PNP_ADMM_CNC_D', 4 decimals, alpha
    in the average line), S6:536 ('PNP_ADMM_CNC_DnCNN')."""
    P, S, D = env['P'], env['S'], env['D']
    mask = golden_inputs['masks']['Q_Random30'].astype(np.float64)
    noises = golden_inputs['noises']
    ts, res = _testset(tmp_path, golden_inputs['gray'])
    kw = dict(testsets=ts, results=str(res))
    P.ADMM_L1(mask, noises, iter_num=3, lambda1=0.1, reo=0.015, **kw)
    P.ADMM_CNC(mask, noises, alpha=0.45, iter_num=3, lambda1=0.5, reo=0.05, b=64, **kw)
    sd = D.seeded_state_dict(D.build('dncnn_15')[0], 1)
    S.PNP_ADMM_L1_D('dncnn_15', mask, noises, model=sd, iter_num=2, reo=0.15, **kw)
    S.PNP_ADMM_CNC_D('dncnn_15', mask, noises, model=sd, alpha=0.9, iter_num=2, lambda1=1.0, reo=0.45, b=0.3, **kw)
    S.PNP_ADMM_CNC_DnCNN('dncnn_25', 'dncnn_15', mask, noises, model=sd, alpha=1.2, iter_num=2, lambda1=4, reo=0.45, b=0.3, **kw)
    want = {
        'Set1_dn_ADMM_L1': ('05_PDG L1.png', 2, None),
        'Set1_dn_ADMM_CNC': ('05_ADMM CNC.png', 4, None),
        'Set1_dn_dncnn_15': (None, None, None),                    # two solvers share this directory (as in the reference)
        'Set1_dn_dncnn_25_dncnn_15': ('05PNP_ADMM_CNC_DnCNN.png', 4, 'alpha'),
    }
    assert sorted(p.name for p in res.iterdir()) == sorted(want)
    for d, (png, dec, extra) in want.items():
        log = (res / d / (d + '.log')).read_text()
        if png is None:
            assert sorted(p.name for p in (res / d).glob('*.png')) == ['05PNP_ADMM_CNC_D.png', '05_dncnn_15_PNP_ADMM_L1_D.png']
            assert re.search(r'05\.png - PSNR: \d+\.\d{2} dB; SSIM: -?\d\.\d{4} ; RE: \d+\.\d{4}\.', log)      # S3:320
            assert re.search(r'05\.png - PSNR: \d+\.\d{4} dB; SSIM: -?\d\.\d{4} ; RE: \d+\.\d{4}\.', log)      # S6:332
            assert re.search(r'testset_name: \(Set1\), alpha: \(0\.900\), Average PSNR:', log)                  # S6:345
            assert re.search(r'testset_name: \(Set1\), Average PSNR:', log)                                     # S3:332
            continue
        assert [p.name for p in (res / d).glob('*.png')] == [png]
        assert re.search(r'05\.png - PSNR: \d+\.\d{%d} dB; SSIM: -?\d\.\d{4} ; RE: \d+\.\d{4}\.' % dec, log)
        assert not re.search(r'05\.png - PSNR: \d+\.\d{%d} dB' % (6 - dec), log)
        assert ('alpha: (1.200)' in log) == (extra == 'alpha')


def test_mat_inputs_through_the_entry_point(env, golden_inputs, golden_admm, tmp_path):
    """The reference's whole input path (S4:182-191 + S4:83-94): masks and noises from CS_MRI/*.mat
    (committed fixture), image from testsets/Set1 -- result equals the golden x of the unmodified script."""
    from pnp_admm_cnc_mri_amd import imageio as IO
    P = env['P']
    mask, noises = IO.load_cs_mri(os.path.join(GOLD, 'cs_mri_fixture'))
    ts, res = _testset(tmp_path, golden_inputs['gray'])
    out = P.ADMM_L1(mask[0], noises, testsets=ts, results=str(res), **P.PRESETS['ADMM_L1'])
    assert rel_l2(out[0], golden_admm['l1_random30_it50']) <= 1e-5
    out = P.ADMM_CNC(mask[2], noises, testsets=ts, results=str(res), **P.PRESETS['ADMM_CNC'])
    assert rel_l2(out[0], golden_admm['cnc_cartesian30_it50']) <= 1e-4        # fp32 CNC at 50 iterations


# ----------------------------------------------------------------------------------------------
# edge behaviour
# ----------------------------------------------------------------------------------------------
def test_zero_iterations_return_the_initial_x(env, golden_inputs):
    """iter_num = 0: the reference's loop body never runs and `out` holds x = |ifft2(y)| (S4:103, 138)."""
    P, S, D = env['P'], env['S'], env['D']
    mask = golden_inputs['masks']['Q_Radial30'].astype(np.float64)
    img = golden_inputs['gray'][None]
    y = O.synthesize(O.requantise(golden_inputs['gray']), mask, golden_inputs['noises'])
    x0 = np.abs(np.fft.ifft2(y))
    for out in (P.ADMM_L1(mask, golden_inputs['noises'], images=img, iter_num=0),
                P.ADMM_CNC(mask, golden_inputs['noises'], images=img, iter_num=0),
                S.PNP_ADMM_L1_D('dncnn_15', mask, golden_inputs['noises'], images=img, iter_num=0,
                                model=D.seeded_state_dict(D.build('dncnn_15')[0], 1)),
                S.PNP_ADMM_CNC_D('dncnn_15', mask, golden_inputs['noises'], images=img, iter_num=0,
                                 model=D.seeded_state_dict(D.build('dncnn_15')[0], 1))[0]):
        assert rel_l2(out[0], x0) <= 2e-6


@pytest.mark.parametrize('H', [256, 512])
def test_unsampled_measurements_never_reach_the_result(env, H):
    """y entries where the mask is 0 are not part of the problem (S4:121-122 reads y[index] only):
    NaN / Inf there must not poison the fused path (Hermitian tables select, they do not multiply)."""
    P = env['P']
    rng = np.random.default_rng(5)
    mask = (rng.uniform(size=(H, H)) < 0.3).astype(np.uint8)
    mask[0, 0] = 1
    B = 3
    img = np.stack([O.phantom(b, H, H) for b in range(B)])
    y = np.stack([O.synthesize(img[b], mask, O.kspace_noise(b, H, H)) for b in range(B)]).astype(np.complex64)
    bad = y.copy()
    hole = (mask == 0)
    bad[:, hole] = np.where(rng.uniform(size=int(hole.sum())) < 0.5, np.nan, np.inf).astype(np.float32) * (1 + 1j)
    res = []
    for yy in (y, bad):
        for fast in (1, 0):
            with P.Engine(H, H, Bmax=B) as eng:
                eng.set_fast_path(fast)
                eng.upload(yy, mask)
                eng.set_state(np.abs(np.fft.ifft2(y * mask)).astype(np.float32), np.zeros((B, H, H), np.float32))
                eng.admm_cnc(3, 0.45, 0.5, 0.05, 64)
                res.append(eng.x())
    assert all(np.isfinite(r).all() for r in res)
    assert np.array_equal(res[0], res[2]) and np.array_equal(res[1], res[3])


def test_device_mask_ids_are_validated(env):
    """mask_id handed over as a DEVICE array is range-checked like a host array (it indexes the mask bank)."""
    torch, P = env['torch'], env['P']
    from pnp_admm_cnc_mri_amd import _lib
    import ctypes as C
    B, K = 4, 2
    y = torch.zeros((B, 256, 256, 2), dtype=torch.float32, device='cuda')
    bank = torch.ones((K, 256, 256), dtype=torch.uint8, device='cuda')
    with P.Engine(256, 256, Bmax=B) as eng:
        L = _lib.lib()
        good = torch.tensor([0, 1, 1, 0], dtype=torch.int32, device='cuda')
        assert L.pnp_upload_problem(eng._ctx, C.c_void_p(y.data_ptr()), C.c_void_p(bank.data_ptr()), C.c_void_p(good.data_ptr()), B, K, 1) == 0
        for badv in ([0, 2, 0, 0], [0, 0, -1, 0]):
            bad = torch.tensor(badv, dtype=torch.int32, device='cuda')
            rc = L.pnp_upload_problem(eng._ctx, C.c_void_p(y.data_ptr()), C.c_void_p(bank.data_ptr()), C.c_void_p(bad.data_ptr()), B, K, 1)
            assert rc == -1 and b'mask_id' in L.pnp_last_error()
            assert L.pnp_init_state(eng._ctx) == -3                      # no problem is left behind
        sch = eng.schedule
        assert set(sch) == {'queues', 'mixed', 'chunk'}


# ----------------------------------------------------------------------------------------------
# PnP coverage
# ----------------------------------------------------------------------------------------------
def _ircnn_bank(D, gain=0.5):
    net, nlm, _ = D.build('ircnn_gray')
    return {str(k): D.seeded_state_dict(net, 300 + k, gain=gain) for k in range(25)}, nlm


@pytest.mark.parametrize('solver', ['cnc', 'l1'])
def test_ircnn_bank_switches_like_the_reference(env, golden_inputs, solver, tmp_path):
    """IRCNN's 25-model bank (S6:184-197): the index ceil(sigma_i * 255 / 2) - 1 changes along the
    sigma schedule and the weights are reloaded on change (S6:289-298, S3:53-62 for the forward).  The
    reference branch itself cannot run under NumPy 2 (`np.int`), so parity is against the oracle loop
    driven with the same GPU denoiser and the same switch; 12 iterations walk through 9 bank entries."""
    torch, D, S = env['torch'], env['D'], env['S']
    from pnp_admm_cnc_mri_amd import utils_pnp
    bank, nlm = _ircnn_bank(D)
    iters = 12
    mask = golden_inputs['masks']['Q_Random30'].astype(np.float64)
    y = O.synthesize(O.requantise(golden_inputs['gray']), mask, golden_inputs['noises'])
    sig = utils_pnp.get_rho_sigma(max(0.255 / 255., nlm), iters, 49, nlm * 255., 1.0)[1]
    idx = [int(np.ceil(float(s) * 255. / 2.) - 1) for s in sig]
    assert idx[0] == 24 and idx[-1] == 7 and len(set(idx)) >= 8
    net, _, _ = D.build('ircnn_gray')
    net.load_state_dict(bank['0'])
    den = D.Denoiser('ircnn_gray', net.eval(), nlm, sigmas=torch.tensor(sig), bank=bank).to(torch.device('cuda'))
    loaded = []

    def denoise(a, i):
        den.select_bank(i)
        loaded.append(den.former_idx)
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))[None, None].cuda()
        return den(t, i)[0, 0].cpu().numpy()

    if solver == 'cnc':
        opts = dict(alpha=0.5, lambda1=1.3, reo=0.45, b=2)                           # S6:575 preset
        out, _ = S.PNP_ADMM_CNC_D('ircnn_gray', mask, None, y=y[None], model=bank, results=str(tmp_path), iter_num=iters, **opts)
        ref = O.pnp_admm_cnc(y, mask, denoise, iters, opts['alpha'], opts['lambda1'], opts['reo'], opts['b'])
    else:
        out = S.PNP_ADMM_L1_D('ircnn_gray', mask, None, y=y[None], model=bank, results=str(tmp_path), iter_num=iters, reo=0.145)   # S3:345
        ref = O.pnp_admm_l1(y, mask, denoise, iters, 0.145)
    assert sorted(set(loaded)) == sorted(set(idx))
    assert rel_l2(out[0], ref) <= 1e-5, rel_l2(out[0], ref)
    # the switch matters: a run pinned to bank entry 0 gives a different image
    pinned = (S.PNP_ADMM_CNC_D('ircnn_gray', mask, None, y=y[None], model=bank['0'], results=str(tmp_path), iter_num=iters, **opts)[0]
              if solver == 'cnc' else
              S.PNP_ADMM_L1_D('ircnn_gray', mask, None, y=y[None], model=bank['0'], results=str(tmp_path), iter_num=iters, reo=0.145))
    assert rel_l2(pinned[0], ref) > 1e-3


def test_x8_all_eight_modes_against_the_reference(env, golden_inputs, tmp_path):
    """PNP_ADMM_L1_D('drunet_gray') for 9 iterations: the x8 cycle i % 8 visits every view, including
    modes 3 and 5 whose inverse is 8 - i % 8 (S3:40-50).  (1) golden x of the unmodified S3 script
    (CPU conv vs MIOpen conv: 2e-4 as for the 3-iteration fixtures); (2) oracle loop with the same GPU
    denoiser <= 1e-5; (3) swapping the inverse of modes 3/5 is caught."""
    torch, D, S = env['torch'], env['D'], env['S']
    from pnp_admm_cnc_mri_amd import utils_pnp
    name = 'drunet_gray'
    opts = dict(env['known']['l1_d_drunet_gray_it9_opts'])
    iters = int(opts.pop('iter_num'))
    assert iters == 9
    mask = golden_inputs['masks']['Q_Random30'].astype(np.float64)
    net, nlm, _ = D.build(name)
    sd = D.seeded_state_dict(net, env['known']['seeds'][name])
    out = S.PNP_ADMM_L1_D(name, mask, golden_inputs['noises'], images=golden_inputs['gray'][None], model=sd,
                          results=str(tmp_path), iter_num=iters, **opts)
    ref = env['gold']['l1_d_drunet_gray_it9']
    assert rel_l2(out[0], ref) <= 2e-4, rel_l2(out[0], ref)

    net.load_state_dict(sd)
    sig = torch.tensor(utils_pnp.get_rho_sigma(max(0.255 / 255., nlm), iters, 49, nlm * 255., 1.0)[1])
    den = D.Denoiser(name, net.eval(), nlm, sigmas=sig, x8=True).to(torch.device('cuda'))

    def denoise(a, i):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))[None, None].cuda()
        return den(t, i)[0, 0].cpu().numpy()

    # identical inputs on both sides: the same complex64 measurements go to the engine (y=) and, widened, to the oracle loop
    # (nine passes through a random-weight U-Net amplify the 1e-7 differences of two float32 syntheses beyond 1e-5)
    y = O.synthesize(O.requantise(golden_inputs['gray']), mask, golden_inputs['noises']).astype(np.complex64)
    out_y = S.PNP_ADMM_L1_D(name, mask, None, y=y[None], model=sd, results=str(tmp_path), iter_num=iters, **opts)
    refo = O.pnp_admm_l1(y.astype(np.complex128), mask, denoise, iters, opts['reo'])
    assert rel_l2(out_y[0], refo) <= 1e-5, rel_l2(out_y[0], refo)

    class WrongInverse(D.Denoiser):
        def _one(self, x, i):
            x = D.augment_img_tensor4(x, i % 8)
            s = self.sigmas[i].float().reshape(1, 1, 1, 1).expand(x.shape[0], 1, x.shape[2], x.shape[3])
            x = D.test_mode(self.model, torch.cat((x, s), dim=1), mode=2, refield=32, min_size=256, modulo=16)
            return D.augment_img_tensor4(x, i % 8)                    # modes 3 and 5 NOT inverted by their partner
    bad = WrongInverse(name, net, nlm, sigmas=sig, x8=True).to(torch.device('cuda'))

    def denoise_bad(a, i):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))[None, None].cuda()
        return bad(t, i)[0, 0].cpu().numpy()
    assert rel_l2(O.pnp_admm_l1(y, mask, denoise_bad, iters, opts['reo']), ref) > 1e-2


def test_config3_full_batch_properties(env, monkeypatch):
    """Config 3 at full size: PNP_ADMM_CNC_D with FFDNet on 512 slices of 256x256, Q_Radial30, S6:573
    preset, 2 iterations.  Size-independent properties: (a) a 64-slice sub-batch run alone is
    bit-equal to the same slices inside the full batch; (b) slices are independent -- permuting the
    batch permutes the result; (c) oracle-loop spot check on the first and the last slice."""
    torch, D, S = env['torch'], env['D'], env['S']
    # the module fixture's deterministic flag narrows MIOpen's choice of kernels -- for this stack at 64 images per call to one
    # that is two orders of magnitude slower on a fresh box (113 s for this test against 8 s; see Denoiser.miopen_find).  The
    # assertions below hold without it (the small-batch tests of this module keep it: run-to-run differences of the faster
    # kernels, amplified by 9 PnP iterations, exceed their 1e-5).
    monkeypatch.setattr(torch.backends.cudnn, 'deterministic', False)
    from pnp_admm_cnc_mri_amd import synthetic as SY
    B = 512
    mask = SY.reference_masks()['Q_Radial30'].astype(np.uint8)
    with env['P'].Engine(256, 256, Bmax=B) as eng:                               # measurements on the device
        img, noise = SY.batch(0, B)
        eng.synthesize(img, noise, mask)
        ys = eng.download_y()
    name = 'ffdnet_gray'
    net, nlm, _ = D.build(name)
    sd = D.seeded_state_dict(net, 1001)
    opts = dict(alpha=0.9, iter_num=2, lambda1=1.35, reo=0.45, b=0.3)            # S6:573, 2 iterations
    full, _ = S.PNP_ADMM_CNC_D(name, mask, None, y=ys, model=sd, **opts)
    full = np.stack(full[:B])
    assert full.shape == (B, 256, 256) and np.isfinite(full).all() and full.min() >= 0 and full.max() <= 1
    sub, _ = S.PNP_ADMM_CNC_D(name, mask, None, y=ys[128:192], model=sd, **opts)
    assert np.array_equal(np.stack(sub[:64]), full[128:192])
    perm = np.random.default_rng(0).permutation(B)
    pm, _ = S.PNP_ADMM_CNC_D(name, mask, None, y=ys[perm], model=sd, **opts)
    pm = np.stack(pm[:B])
    assert np.abs(pm - full[perm]).max() <= 5e-6                                # conv batches regroup: float32 rounding of a 15-layer stack only
    net.load_state_dict(sd)
    den = D.Denoiser(name, net.eval(), nlm).to(torch.device('cuda'))

    def denoise(a, i):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))[None, None].cuda()
        return den(t, i)[0, 0].cpu().numpy()
    for b in (0, B - 1):
        ref = O.pnp_admm_cnc(ys[b].astype(np.complex128), mask, denoise, 2, 0.9, 1.35, 0.45, 0.3)
        assert rel_l2(full[b], ref) <= 1e-5, (b, rel_l2(full[b], ref))
    # (d) the split-half f16 backend at the same size: every one of the 512 slices within 2e-5 of the PyTorch / MIOpen run, the
    # same in-range values, the 64-slice sub-batch bit-equal to its slices inside the batch, and an oracle-loop spot check driven
    # by the f16x3 denoiser itself
    fh, _ = S.PNP_ADMM_CNC_D(name, mask, None, y=ys, model=sd, cnn_backend='hip_f16x3', **opts)
    fh = np.stack(fh[:B])
    per_slice = np.linalg.norm((fh - full).reshape(B, -1), axis=1) / np.linalg.norm(full.reshape(B, -1), axis=1)
    assert np.isfinite(fh).all() and fh.min() >= 0 and fh.max() <= 1 and per_slice.max() <= 2e-5, per_slice.max()
    subh, _ = S.PNP_ADMM_CNC_D(name, mask, None, y=ys[128:192], model=sd, cnn_backend='hip_f16x3', **opts)
    assert np.array_equal(np.stack(subh[:64]), fh[128:192])
    denh = D.Denoiser(name, net, nlm, backend='hip_f16x3').to(torch.device('cuda'))

    def denoise_h(a, i):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))[None, None].cuda()
        return denh(t, i)[0, 0].cpu().numpy()
    ref = O.pnp_admm_cnc(ys[B - 1].astype(np.complex128), mask, denoise_h, 2, 0.9, 1.35, 0.45, 0.3)
    assert rel_l2(fh[B - 1], ref) <= 1e-5, rel_l2(fh[B - 1], ref)


# ----------------------------------------------------------------------------------------------
# the float instance of the split-chain column kernel (PNP_FUSED_COLS=2)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize('solver', ['cnc', 'l1'])
def test_split_chain_float_engine_vs_oracle(env, golden_inputs, solver, monkeypatch):
    """k_fcols2<float> (one column chain per thread and slice, partner slice in lane ^ 1) behind the same
    C ABI: 5 slices (odd: the last pair holds one), three masks, 10 iterations against the oracle, and
    the teacher-forced x-update against the default engine."""
    P = env['P']
    masks = np.stack([golden_inputs['masks'][k] for k in ('Q_Random30', 'Q_Radial30', 'Q_Cartesian30')]).astype(np.uint8)
    B = 5
    mid = (np.arange(B) % 3).astype(np.int32)
    ys = np.stack([O.synthetic_problem(b, masks[mid[b]])[1] for b in range(B)]).astype(np.complex64)
    monkeypatch.setenv('PNP_FUSED_COLS', '2')
    with P.Engine(256, 256, Bmax=B) as eng:
        eng.upload(ys, masks, mid)
        assert eng.path_name == 'fused'
        eng.init_state()
        if solver == 'cnc':
            eng.admm_cnc(10, 0.45, 0.5, 0.05, 64)
        else:
            eng.admm_l1(10, 0.1, 0.015)
        x = eng.x()
    for b in range(B):
        y128 = ys[b].astype(np.complex128)
        ref = O.admm_cnc(y128, masks[mid[b]], 10) if solver == 'cnc' else O.admm_l1(y128, masks[mid[b]], 10)
        assert rel_l2(x[b], ref) <= 2e-6, (b, rel_l2(x[b], ref))


def test_config4_shard_full_batch_drunet(env, monkeypatch):
    """Config 4's per-GPU shard at full size: PNP_ADMM_CNC_D with DRUNet on 512 slices of 256x256, Q_Cartesian30,
    S6:577 preset, ONE iteration (two DRUNet forwards over the whole shard) on the split-half f16 backend -- every convolution of the
    U-Net on libpnpmri.so.  (a) the 64 slices of one CNN batch run alone are bit-equal to the same slices inside the shard; (b) the
    oracle loop driven by the same denoiser on the last slice <= 1e-5; (c) the PyTorch / MIOpen backend on the shard's last 8 slices (round
    5 ran it on all 512: 105 s of the suite, nearly all of it MIOpen compiling and searching batch-64 shapes on a fresh box; the per-slice
    comparison does not get stronger with more slices of the same shape): every slice within 2e-5 of the f16x3 run, the last <= 1e-5 from
    the oracle loop driven by the MIOpen denoiser."""
    torch, D, S = env['torch'], env['D'], env['S']
    from pnp_admm_cnc_mri_amd import synthetic as SY, utils_pnp
    monkeypatch.setattr(torch.backends.cudnn, 'deterministic', False)           # as in test_config3_full_batch_properties
    B = 512
    mask = SY.reference_masks()['Q_Cartesian30'].astype(np.uint8)
    with env['P'].Engine(256, 256, Bmax=B) as eng:
        img, noise = SY.batch(0, B)
        eng.synthesize(img, noise, mask)
        ys = eng.download_y()
    name = 'drunet_gray'
    net, nlm, _ = D.build(name)
    sd = D.seeded_state_dict(net, 1003)
    opts = dict(alpha=1, iter_num=1, lambda1=0.8, reo=0.8, b=0.45)               # S6:577, 1 iteration
    fh, _ = S.PNP_ADMM_CNC_D(name, mask, None, y=ys, model=sd, cnn_backend='hip_f16x3', **opts)
    fh = np.stack(fh[:B])
    assert np.isfinite(fh).all() and fh.min() >= 0 and fh.max() <= 1
    subh, _ = S.PNP_ADMM_CNC_D(name, mask, None, y=ys[448:512], model=sd, cnn_backend='hip_f16x3', **opts)
    assert np.array_equal(np.stack(subh[:64]), fh[448:512])
    net.load_state_dict(sd)
    sig = torch.tensor(utils_pnp.get_rho_sigma(max(0.255 / 255., nlm), 1, 49, nlm * 255., 1.0)[1])

    def oracle_last(backend):
        den = D.Denoiser(name, net.eval(), nlm, sigmas=sig, backend=backend).to(torch.device('cuda'))

        def denoise(a, i):
            t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))[None, None].cuda()
            return den(t, i)[0, 0].cpu().numpy()
        return O.pnp_admm_cnc(ys[B - 1].astype(np.complex128), mask, denoise, 1, 1, 0.8, 0.8, 0.45)
    assert rel_l2(fh[B - 1], oracle_last('hip_f16x3')) <= 1e-5
    # the PyTorch / MIOpen backend on the shard's last 8 slices, one CNN call of 8 (below MIOpen's find threshold of 16 images per call: at
    # 64 per call a fresh box spends ~100 s compiling and searching DRUNet's convolution shapes -- rounds 2-5 paid that here)
    full, _ = S.PNP_ADMM_CNC_D(name, mask, None, y=ys[504:512], model=sd, cnn_backend='torch', cnn_batch=8, **opts)
    full = np.stack(full[:8])
    per_slice = np.linalg.norm((fh[504:512] - full).reshape(8, -1), axis=1) / np.linalg.norm(full.reshape(8, -1), axis=1)
    assert np.isfinite(full).all() and full.min() >= 0 and full.max() <= 1 and per_slice.max() <= 2e-5, per_slice.max()
    assert rel_l2(full[7], oracle_last('torch')) <= 1e-5


# ----------------------------------------------------------------------------------------------
# chunked schedules (defaults of the 512x512 and of the double-precision 256x256 loops)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize('H,precision,B,iters', [(512, 'f32', 52, 4), (256, 'f64', 100, 4)])
def test_chunked_default_schedules_are_bit_identical(env, H, precision, B, iters):
    """512x512 loops run all iterations on 48 slices before the next 48, the double 256x256 engine on 96 (a chunk's working
    set stays in the Infinity Cache): pure scheduling -- the same bits as the whole batch at once (chunk < 0) and as any
    other chunk size, for x and for the state."""
    P = env['P']
    from pnp_admm_cnc_mri_amd import synthetic as S
    masks = np.stack([S.synthetic_mask(k, H, H) for k in ('random', 'radial', 'cartesian')]).astype(np.uint8)
    img, noise = S.batch(0, B, H, H)
    mid = np.arange(B) % 3
    with P.Engine(H, H, Bmax=B) as e32:                      # measurements from the float engine (the double one has no synthesis)
        e32.synthesize(img, noise, masks, mid)
        y = e32.download_y()
    res = []
    with P.Engine(H, H, Bmax=B, precision=precision) as eng:
        for chunk in (0, -1, 10):
            eng.set_schedule(queues=2, mixed_launches=True, chunk=chunk)
            assert eng.schedule['chunk'] == chunk
            eng.upload(y, masks, mid)
            assert eng.path_name == 'fused'
            eng.init_state()
            eng.admm_cnc(iters, 0.45, 0.5, 0.05, 64)
            x = eng.x()
            z, w = eng.get_state()
            eng.init_state()
            eng.admm_l1(iters, 0.1, 0.015)
            res.append((x, z, w, eng.x()))
    assert np.isfinite(res[0][0]).all()
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert np.array_equal(a, b)


def test_plan_reports_what_the_loops_will_do(env):
    """pnp_get_plan: queues / slices per chunk / launches per iteration of the next run, per path."""
    P = env['P']
    mask = np.ones((256, 256), np.uint8)
    with P.Engine(256, 256, Bmax=512) as eng:
        eng.upload(np.zeros((512, 256, 256), np.complex64), mask)
        assert eng.path_name == 'slice' and eng.plan == {'queues': 1, 'chunk': 512, 'launches_per_iteration': 0}
        eng.upload(np.zeros((40, 256, 256), np.complex64), mask)
        assert eng.path_name == 'fused' and eng.plan['launches_per_iteration'] == 2 * eng.plan['queues']
    with P.Engine(512, 512, Bmax=64) as eng:
        eng.upload(np.zeros((52, 512, 512), np.complex64), np.ones((512, 512), np.uint8))
        assert eng.plan == {'queues': 4, 'chunk': 16, 'launches_per_iteration': 8}
        eng.set_schedule(queues=1, mixed_launches=False, chunk=0)
        assert eng.plan == {'queues': 1, 'chunk': 48, 'launches_per_iteration': 4}
        eng.set_schedule(queues=2, mixed_launches=False, chunk=-1)
        assert eng.plan == {'queues': 1, 'chunk': 52, 'launches_per_iteration': 2}
    with P.Engine(256, 256, Bmax=100, precision='f64') as eng:
        eng.upload(np.zeros((100, 256, 256), np.complex64), mask)
        assert eng.plan == {'queues': 4, 'chunk': 24, 'launches_per_iteration': 10}


def test_bench_runs_its_collectives_on_rccl_with_one_rank():
    """The N > 1 code path of bench.py on the real backend: PNP_BENCH_FORCE_DIST=1 makes a one-rank job initialise
    the 'nccl' (= RCCL) process group and run the barriers, the device-tensor gather and the MAX all-reduce.  Same
    slices, same iterations as the plain run: the checksum must be equal and the line complete."""
    env_ = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    base = [sys.executable, os.path.join(ROOT, 'bench.py'), '--batch', '64', '--steps', '3', '--warmup', '1', '--no-cpu-baseline',
            '--sustain-s', '0.2']
    r = subprocess.run(base, env=dict(env_, PNP_BENCH_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port)),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    # the contract is ONE line on stdout: RCCL's version banner (it prints one when the process group comes up) must not be there
    assert len(r.stdout.decode().strip().splitlines()) == 1, r.stdout.decode()[:400]
    j = json.loads(r.stdout.decode().strip())
    assert j['n_gpus'] == 1 and j['gather_ms'] is not None and j['gather_ms'] > 0 and j['x_finite']
    assert j['config']['path'] == 'slice' and len(j['per_rank']['ms_per_step']) == 1
    r1 = subprocess.run(base, env=env_, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r1.returncode == 0, r1.stderr.decode()[-3000:]
    j1 = json.loads([l for l in r1.stdout.decode().splitlines() if l.startswith('{')][0])
    assert j1['gather_ms'] is None and j1['x_checksum'] == j['x_checksum']


def test_bench_line_carries_parity_and_the_f64_record():
    """With the CPU legs on, rank 0 of an N = 1 job re-proves parity of the timed run against the oracle (three slices,
    W + K iterations) and times the double-precision engine in the same process."""
    env_ = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--batch', '64', '--steps', '8', '--warmup', '2',
                        '--cpu-budget', '1', '--sustain-s', '0.5', '--no-pnp-record'], env=env_, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    j = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith('{')][0])
    # (the `pnp` sub-record has its own test, test_bench_pnp_sub_record)
    # round 6: the line is attributable -- which card, its nominal clock, and what its memory system gives a streaming kernel with the slice
    # kernel's access shape, measured >= 0.6 s before the timed region
    dv, cal = j['config']['device'], j['roofline']['calibration']
    assert dv['arch'].startswith('gfx950') and dv['compute_units'] == 256 and dv['clock_mhz'] > 1000 and ':' in dv['pci_bus_id'], dv
    assert 3000 < cal['calibration_gbs'] < 8000 and cal['slices'] == 64, cal
    assert abs(j['roofline']['frac_of_calibration'] - j['roofline']['achieved'] / cal['calibration_gbs']) <= 1e-12
    # the reference's own usage: one 256 x 256 slice, the committed 50 iterations, ms per solve, each against the oracle
    lat = j['latency']
    assert 'error' not in lat, lat
    assert lat['admm_l1_f32']['rel_l2_vs_oracle'] <= 1e-5 and lat['admm_cnc_f64']['rel_l2_vs_oracle'] <= 1e-5 and lat['admm_cnc_f32']['rel_l2_vs_oracle'] <= 1e-4, lat
    for k in ('pnp_cnc_d_ffdnet_gray_hip_f16x3', 'pnp_cnc_d_drunet_gray_hip_f16x3'):
        assert lat[k]['rel_l2_vs_oracle'] <= 1e-5 and 0 < lat[k + '_graph']['ms'], (k, lat[k], lat[k + '_graph'])
    assert 0 < lat['admm_cnc_f32']['ms'] < 50 and 0 < lat['pnp_cnc_d_ffdnet_gray_hip_f16x3']['ms'] < 500, lat
    assert j['parity']['iterations'] == 10 and len(j['parity']['rel_l2_vs_oracle']) == 3
    assert max(j['parity']['rel_l2_vs_oracle']) <= 1e-5
    assert j['f64']['dtype'] == 'f64' and j['f64']['value'] > 0 and max(j['f64']['rel_l2_vs_oracle']) <= 1e-9
    assert j['cpu_baseline']['cores'] == 1 and j['roofline']['traffic_measured_in_this_run'] is False
    # roofline.frac is a PHYSICAL fraction (bytes the kernels move / time / peak), the 57 N contract figure sits beside it
    rf = j['roofline']
    assert 0 < rf['frac'] <= 1 and abs(rf['frac'] - rf['achieved'] / rf['peak']) <= 1e-12
    assert rf['bytes_from'] in ('pmc', 'own_algorithmic') and rf['frac_contract_57N'] > rf['frac']
    assert rf['own_algorithmic_bytes_per_iteration'] == 20.0 * 65536 * 64 and rf['contract_bytes_per_iteration'] == 57.0 * 65536 * 64
    # the sustained record: >= 0.5 s of back-to-back calls after a pre-heat, same workload; x / parity above are those of W + K iterations
    su = j['sustained']
    assert su['span_s'] >= 0.3 and su['steps'] % 100 == 0 and su['steps_per_call'] == 100 and su['fresh_state_per_call']
    assert su['value'] > 0 and 0 < su['frac'] <= 1
    assert 0.5 * j['value'] <= su['value'] <= 2.0 * j['value']     # an 8-step call pays its prologue and tail; 100-step solves do not


def test_bench_pnp_line_and_its_two_rank_launch():
    """bench_pnp.py (configs 3-5): the line carries the FFT + prox part's bytes / time / HBM fraction and, separately, the
    denoiser's share, FLOP per call and FLOP/s against the fp32 matrix peak; `--gpus 2` launches its own
    ranks like bench.py (both on this box's one GPU, gloo rendezvous) and times the gather apart."""
    env_ = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    base = [sys.executable, os.path.join(ROOT, 'bench_pnp.py'), '--model', 'ffdnet_gray', '--batch', '16', '--steps', '2', '--warmup', '1']
    r = subprocess.run(base, env=env_, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    j = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith('{')][0])
    assert j['n_gpus'] == 1 and j['gather_ms'] is None and j['x_finite'] and j['denoiser_outputs_finite']
    assert j['config']['weights'] == 'trained' and j['parity']['iterations'] == 3 and j['parity']['slices'] == [0, 7, 15]      # ffdnet_gray: the trained fixture network
    assert max(j['parity']['rel_l2_vs_oracle']) <= 1e-5, j['parity']
    assert abs(j['denoiser']['flop_per_call_per_slice'] / 1e9 - 15.9) <= 0.4
    assert j['denoiser']['roofline']['bound'] == 'mfma_f32' and 0 < j['denoiser']['roofline']['frac'] < 1
    assert j['fft_prox']['roofline']['bound'] == 'hbm' and j['fft_prox']['algorithmic_bytes'] == 57.0 * 65536 * 16
    assert abs(j['denoiser']['share'] + j['fft_prox']['share'] - 1) < 1e-9
    # four ranks (the box allows six processes on its card: this one + four ranks), the f16x3 backend, the He-scaled weights of the earlier rounds
    r2 = subprocess.run(base[:4] + ['--batch', '8', '--steps', '2', '--warmup', '1', '--gpus', '4', '--rehearse-gloo', '--cnn-backend', 'hip_f16x3',
                                    '--weights', 'he'], env=env_, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r2.returncode == 0, r2.stderr.decode()[-3000:]
    lines = [l for l in r2.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    j2 = json.loads(lines[0])
    assert j2['n_gpus'] == 4 and j2['gather_ms'] is not None and j2['scaling'] == 'weak' and j2['x_finite'] and j2['parity'] is None
    assert len(j2['per_rank']['ms_per_step']) == 4 and abs(max(j2['per_rank']['ms_per_step']) - j2['ms_per_step']) <= 1e-9
    assert j2['config']['weights'] == 'he' and j2['denoiser']['roofline']['bound'] == 'mfma_f16' and 0 < j2['denoiser']['roofline']['frac'] <= 1
    # the N > 1 path on the real backend: one rank, 'nccl' (= RCCL) process group, device-tensor gather, all_gather of the clocks
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    r3 = subprocess.run(base, env=dict(env_, PNP_BENCH_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port)),
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r3.returncode == 0, r3.stderr.decode()[-3000:]
    assert len(r3.stdout.decode().strip().splitlines()) == 1, r3.stdout.decode()[:400]      # RCCL's banner must not reach stdout
    j3 = json.loads(r3.stdout.decode().strip())
    assert j3['n_gpus'] == 1 and j3['gather_ms'] is not None and j3['gather_ms'] > 0 and len(j3['per_rank']['ms_per_step']) == 1


def test_bench_line_at_512_with_its_cpu_legs():
    """`bench.py --size 512` (the shape of config 5) with the checker legs on: the metric string names the shape, the oracle
    comparison and both CPU baselines run on 512 x 512 data (they once built 256 x 256 phantoms for a 512 x 512 mask)."""
    env_ = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--size', '512', '--batch', '8', '--steps', '2', '--warmup', '1',
                        '--cpu-budget', '0.5', '--sustain-s', '0.2'], env=env_, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    j = json.loads(r.stdout.decode().strip())
    assert '512x512' in j['metric'] and j['config']['path'] == 'fused' and j['f64'] is None
    assert max(j['parity']['rel_l2_vs_oracle']) <= 1e-5 and j['parity']['iterations'] == 3
    assert j['cpu_baseline']['value'] > 0 and j['cpu_baseline_all_cores'].get('value', 0) > 0


def test_example_driver_runs_the_pnp_mains_like_the_reference(golden_inputs, tmp_path):
    """examples/run_reference_defaults.py in its PnP form = what `python "【6】PNP_ADMM_CNC_D .py"` does (S6:569-620): a reference-shaped tree
    (CS_MRI/*.mat, testsets/Set1/05.png, model_zoo/<model>.pth with KAIR keys), the committed preset, the authors' log line -- here with the
    contractive fixture weights, whose 50-iteration result the unmodified script produced too (tests/golden/pnp50_set1_05.npz)."""
    import torch
    from conftest import weights50
    ts, res = _testset(tmp_path, golden_inputs['gray'])
    os.symlink(os.path.join(GOLD, 'cs_mri_fixture'), tmp_path / 'CS_MRI')
    (tmp_path / 'model_zoo').mkdir()
    torch.save(weights50('ffdnet_gray'), tmp_path / 'model_zoo' / 'ffdnet_gray.pth')
    env_ = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'examples', 'run_reference_defaults.py'), '--root', str(tmp_path), '--solver', 'pnp_cnc',
                        '--model', 'ffdnet_gray', '--cnn-backend', 'hip_f16x3', '--results', str(res)], env=env_, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    out = r.stdout.decode()
    known = json.load(open(os.path.join(GOLD, 'pnp_known.json')))['known50']['cnc_d_ffdnet_gray']           # '05.png - PSNR: 16.3045 dB; ...'
    want = float(known.split('PSNR:')[1].split('dB')[0])
    got = float([l for l in out.splitlines() if l.startswith('psnr')][0].split("'")[1])
    assert abs(got - want) <= 0.01, (got, known)
    log = (res / 'Set1_dn_ffdnet_gray' / 'Set1_dn_ffdnet_gray.log').read_text()
    assert 'PSNR: %.4f dB' % got in log and '05PNP_ADMM_CNC_D.png' in os.listdir(res / 'Set1_dn_ffdnet_gray')
