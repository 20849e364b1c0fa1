#!/usr/bin/env python3
"""Golden vectors for the PnP entry points, from the UNMODIFIED reference scripts
("【6】PNP_ADMM_CNC_D .py", "【3】PNP_ADMM_L1_D  .py") run on CPU in the build container.

TEST INFRASTRUCTURE (see oracle/make_golden.py for how the scripts are run).  The reference ships
no weights (model_zoo/README.md), so the nets get deterministic synthetic weights from
pnp_admm_cnc_mri_amd.denoisers.seeded_state_dict -- saved as model_zoo/<name>.pth in the scratch
dir and loaded by the reference's own model classes with strict=True, which also proves that the
build's module declarations have KAIR's exact state_dict keys and shapes.

ircnn_gray cannot be generated: the reference's bank switch uses `np.int` (S6:290), removed in
NumPy >= 1.24, so that branch raises under the NumPy 2.2.6 of this image.
"""
import contextlib
import io
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden as MG                                   # noqa: E402
from pnp_admm_cnc_mri_amd import denoisers as D           # noqa: E402

S3 = os.path.join(MG.REF, '【3】PNP_ADMM_L1_D  .py')
S6 = os.path.join(MG.REF, '【6】PNP_ADMM_CNC_D .py')
NAMES = ['drunet_gray', 'ffdnet_gray', 'fdncnn_gray', 'dncnn_15', 'dncnn_25']
SEEDS = {n: 1000 + i for i, n in enumerate(NAMES)}
ITERS = 3


def tiny_net(seed=5):
    """3-layer 3x3 conv net (2 -> 8 -> 8 -> 1 channels, zero padding: border effects depend on where a
    window sits) with NumPy-seeded weights; the split / augmentation goldens use it on both sides."""
    rng = np.random.default_rng(seed)
    net = torch.nn.Sequential(torch.nn.Conv2d(2, 8, 3, 1, 1), torch.nn.ReLU(), torch.nn.Conv2d(8, 8, 3, 1, 1),
                              torch.nn.ReLU(), torch.nn.Conv2d(8, 1, 3, 1, 1))
    with torch.no_grad():
        for p_ in net.parameters():
            p_.copy_(torch.from_numpy(rng.standard_normal(tuple(p_.shape)).astype(np.float32) * 0.3))
    return net.eval()


def extra():
    """Vectors added in round 2, appended to the existing fixture files (nothing else is regenerated):
      * l1_d_drunet_gray_it9 -- S3 with --iter_num 9, so the x8 cycle visits all eight modes
        including the 8 - i%8 inverses of modes 3 and 5 (S3:40-50);
      * split_* / augment_* -- the reference's utils_model.test_split_fn (one- and two-level
        splits) and utils_image.augment_img_tensor4 on small seeded inputs."""
    MG.install_shims()
    d = MG.scratch_dir()
    os.chdir(d)
    os.makedirs('model_zoo')
    n = 'drunet_gray'
    net, _, _ = D.build(n)
    torch.save(D.seeded_state_dict(net, SEEDS[n]), os.path.join('model_zoo', n + '.pth'))
    torch.set_num_threads(8)
    npz = os.path.join(MG.GOLD, 'pnp_set1_05.npz')
    arrays = dict(np.load(npz))
    kj = os.path.join(MG.GOLD, 'pnp_known.json')
    meta = json.load(open(kj))
    known = meta['known']

    g, lines, _ = MG.run_script(S3, ['--iter_num', '9'], 'Set1_dn_drunet_gray')
    arrays['l1_d_drunet_gray_it9'] = np.asarray(g['out'][0], np.float32)
    known['l1_d_drunet_gray_it9_opts'] = {k: float(v) for k, v in g['PNP_ADMM_L1_D_opts5'].items()}
    known['l1_d_drunet_gray_it9_sum'] = float(arrays['l1_d_drunet_gray_it9'].astype(np.float64).sum())
    known['l1_d_drunet_gray_it9_line'] = [l for l in lines if 'PSNR' in l and '05.png' in l][-1]

    sys.path.insert(0, MG.REF)
    from utils import utils_model as ref_um, utils_image as ref_ui
    tn = tiny_net()
    rng = np.random.default_rng(11)
    with torch.no_grad():
        for tag, shape, kw in (('split_1level', (2, 2, 40, 56), dict(refield=8, min_size=32, modulo=4)),
                               ('split_2level', (1, 2, 80, 96), dict(refield=8, min_size=24, modulo=4)),
                               ('split_whole_padded', (1, 2, 30, 27), dict(refield=8, min_size=32, modulo=4))):
            L = torch.from_numpy(rng.random(shape, dtype=np.float32))
            arrays[tag + '_in'] = L.numpy()
            arrays[tag + '_out'] = ref_um.test_mode(tn, L, mode=2, sf=1, **kw).numpy()
            known[tag + '_kw'] = kw
        a = torch.arange(2 * 1 * 5 * 7, dtype=torch.float32).reshape(2, 1, 5, 7)
        for m in range(8):
            arrays['augment_mode%d' % m] = ref_ui.augment_img_tensor4(a, m).contiguous().numpy()
    sys.path.remove(MG.REF)

    np.savez_compressed(npz, **arrays)
    with open(kj, 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print({k: v.shape for k, v in arrays.items()})


SEEDS50 = {'drunet_gray': 2000, 'ffdnet_gray': 2001, 'fdncnn_gray': 2002, 'dncnn_15': 2003, 'dncnn_25': 2004, 'ircnn_gray': 2100}   # ircnn: + bank index


def fifty():
    """Round 5: goldens at the reference's OWN run length -- every PnP preset is 50 iterations (S6:569-577, S3:339-347) -- from the
    unmodified S6 / S3 with CONTRACTIVE seeded weights (denoisers.contractive_state_dict: identity carrier + operator-norm-scaled
    seeded kernels; the operator norms are measured here by power iteration and committed as `gains50`, so that the GPU box builds
    bit-identical weights without measuring anything).  Written to tests/golden/pnp50_set1_05.npz + the 'known50' / 'gains50' /
    'lipschitz50' entries of pnp_known.json; the 3-iteration fixtures stay as they are.

    One more import shim, for the IRCNN branch only: `np.int` (S6:290, S3:281), an alias of the builtin `int` that NumPy 1.24 removed.
    `np.int = int` restores exactly what the reference was written against; the script itself is not touched."""
    import contractive as CT
    MG.install_shims()
    if not hasattr(np, 'int'):
        np.int = int
    d = MG.scratch_dir()
    os.chdir(d)
    os.makedirs('model_zoo')
    torch.set_num_threads(8)
    kj = os.path.join(MG.GOLD, 'pnp_known.json')
    meta = json.load(open(kj))
    gains, lips = {}, {}
    gold_in = np.load(os.path.join(MG.GOLD, 'inputs_set1_05.npz'))
    for n, seed in SEEDS50.items():
        net, nlm, sched = D.build(n)
        fam = D.family(n)
        if fam == 'ircnn':                                    # the 25-model bank, S6:196: {str(index): state_dict}
            bank = {}
            for idx in range(25):
                g = CT.conv_operator_norms(net, seed + idx)
                gains['%s/%d' % (n, idx)] = g
                bank[str(idx)] = D.contractive_state_dict(net, fam, seed + idx, g)
            torch.save(bank, os.path.join('model_zoo', n + '.pth'))
            sd = bank['24']
        else:
            gains[n] = CT.conv_operator_norms(net, seed)
            sd = D.contractive_state_dict(net, fam, seed, gains[n])
            torch.save(sd, os.path.join('model_zoo', n + '.pth'))
        # local Lipschitz constant of the denoiser map at the loop's starting point (recorded, not asserted)
        net.load_state_dict(sd)
        net.eval()
        for p_ in net.parameters():
            p_.requires_grad = False
        sig = torch.tensor(MG_sigmas(nlm, 50)) if sched else None
        den = D.Denoiser(n, net, nlm, sigmas=sig, noises=gold_in['noises_c128'] * 3.0)
        mask0 = np.unpackbits(gold_in['Q_Random30_packbits'])[:65536].reshape(256, 256).astype(np.float64)
        y0 = np.fft.fft2(np.float32(gold_in['gray_u8'] / 255.)) * mask0 + gold_in['noises_c128'] * 3.0
        x0 = torch.from_numpy(np.abs(np.fft.ifft2(y0)).astype(np.float32))[None, None]
        lips[n] = CT.lipschitz_at(lambda t: den._one(t, 0), x0, iters=10)
        print(n, 'Lipschitz at x0 ~ %.4f' % lips[n], flush=True)

    arrays, known = {}, {'numpy': np.__version__, 'torch': torch.__version__, 'iters': 50, 'seeds': SEEDS50,
                         'note': 'x after the presets\' own 50 iterations, 05.png, committed presets, contractive seeded weights; '
                                 'mask Q_Random30 unless the key names another'}

    def line(lines):
        return [l for l in lines if 'PSNR' in l and '05.png' in l][-1]

    def fopts(o):
        return {k: float(v) for k, v in o.items()}

    # ---- S6: its main runs drunet_gray (CNC_D) and the dncnn_25 / dncnn_15 pair at the presets -------------------------------
    g, lines, _ = MG.run_script(S6, [], 'Set1_dn_drunet_gray')
    arrays['cnc_d_drunet_gray'] = np.asarray(g['out1'], np.float32)
    arrays['cnc_dncnn_pair'] = np.asarray(g['out2'], np.float32)
    known['cnc_d_drunet_gray'] = line(lines)
    known['cnc_d_drunet_gray_opts'] = fopts(g['PNP_ADMM_CNC_D_opts4'])
    known['cnc_dncnn_pair_opts'] = fopts(g['PNP_ADMM_CNC_DnCNN_opts'])
    assert int(g['PNP_ADMM_CNC_D_opts4']['iter_num']) == 50
    jobs6 = [('fdncnn_gray', 0, 'PNP_ADMM_CNC_D_opts1', 'cnc_d_fdncnn_gray'), ('ffdnet_gray', 0, 'PNP_ADMM_CNC_D_opts2', 'cnc_d_ffdnet_gray'),
             ('ircnn_gray', 0, 'PNP_ADMM_CNC_D_opts3', 'cnc_d_ircnn_gray'),
             ('ffdnet_gray', 1, 'PNP_ADMM_CNC_D_opts2', 'cnc_d_ffdnet_gray_radial30'),         # BASELINE.json configs[2]: FFDNet, Q_Radial30
             ('drunet_gray', 2, 'PNP_ADMM_CNC_D_opts4', 'cnc_d_drunet_gray_cartesian30')]      # configs[3]: DRUNet, Q_Cartesian30
    for n, k, on, tag in jobs6:
        cap = MG._Capture('Set1_dn_' + n)
        with contextlib.redirect_stdout(io.StringIO()):
            o, _ = g['PNP_ADMM_CNC_D'](n, g['mask'][k], g['noises'], **g[on])
        arrays[tag] = np.asarray(o[0], np.float32)
        known[tag] = line(cap.lines)
        known[tag + '_opts'] = fopts(g[on])
        print(tag, known[tag], flush=True)

    # ---- S3: its main runs drunet_gray (x8 cycle) -----------------------------------------------------------------------------
    g, lines, _ = MG.run_script(S3, [], 'Set1_dn_drunet_gray')
    arrays['l1_d_drunet_gray'] = np.asarray(g['out'][0], np.float32)
    known['l1_d_drunet_gray'] = line(lines)
    known['l1_d_drunet_gray_opts'] = fopts(g['PNP_ADMM_L1_D_opts5'])
    for n, on in (('fdncnn_gray', 'PNP_ADMM_L1_D_opts1'), ('dncnn_15', 'PNP_ADMM_L1_D_opts2'), ('ffdnet_gray', 'PNP_ADMM_L1_D_opts3'),
                  ('ircnn_gray', 'PNP_ADMM_L1_D_opts4')):
        cap = MG._Capture('Set1_dn_' + n)
        with contextlib.redirect_stdout(io.StringIO()):
            o = g['PNP_ADMM_L1_D'](n, g['mask'][0], g['noises'], **g[on])
        arrays['l1_d_' + n] = np.asarray(o[0], np.float32)
        known['l1_d_' + n] = line(cap.lines)
        known['l1_d_' + n + '_opts'] = fopts(g[on])
        print('l1_d_' + n, known['l1_d_' + n], flush=True)

    for k, v in arrays.items():
        assert v.shape == (256, 256) and np.isfinite(v).all(), k
        known[k + '_sum'] = float(v.astype(np.float64).sum())
    np.savez_compressed(os.path.join(MG.GOLD, 'pnp50_set1_05.npz'), **arrays)
    meta['known50'], meta['gains50'], meta['lipschitz50'] = known, gains, lips
    with open(kj, 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps(known, indent=1, sort_keys=True))


def trained():
    """Round 5: 50-iteration goldens of the unmodified S6 / S3 with a TRAINED FFDNet (oracle/train_fixture_denoiser.py ->
    tests/golden/ffdnet_gray_trained.npz): PNP_ADMM_CNC_D on Q_Random30 and Q_Radial30 (BASELINE.json configs[2]'s pair), PNP_ADMM_L1_D on
    Q_Random30, at the committed presets.  The scripts' mains are run with --iter_num 1 on the contractive fixture weights only to obtain
    their solver functions, masks and option dicts; the calls of record pass iter_num = 50 themselves."""
    import contractive as CT
    MG.install_shims()
    if not hasattr(np, 'int'):
        np.int = int
    d = MG.scratch_dir()
    os.chdir(d)
    os.makedirs('model_zoo')
    torch.set_num_threads(8)
    kj = os.path.join(MG.GOLD, 'pnp_known.json')
    meta = json.load(open(kj))
    for n in ('drunet_gray', 'dncnn_25', 'dncnn_15'):               # what the mains load
        net, _, _ = D.build(n)
        torch.save(D.contractive_state_dict(net, D.family(n), meta['known50']['seeds'][n], meta['gains50'][n]), os.path.join('model_zoo', n + '.pth'))
    w = np.load(os.path.join(MG.GOLD, 'ffdnet_gray_trained.npz'))
    net, _, _ = D.build('ffdnet_gray')
    sd = {k: torch.from_numpy(w[k]) for k in net.state_dict()}
    net.load_state_dict(sd, strict=True)
    torch.save(sd, os.path.join('model_zoo', 'ffdnet_gray.pth'))
    npz = os.path.join(MG.GOLD, 'pnp50_set1_05.npz')
    arrays = dict(np.load(npz))
    known = meta['known50']

    def line(lines):
        return [l for l in lines if 'PSNR' in l and '05.png' in l][-1]

    g, _, _ = MG.run_script(S6, ['--iter_num', '1'], 'Set1_dn_drunet_gray')
    for k, tag in ((0, 'trained_cnc_d_ffdnet_gray'), (1, 'trained_cnc_d_ffdnet_gray_radial30')):
        opts = dict(g['PNP_ADMM_CNC_D_opts2'], iter_num=50)
        cap = MG._Capture('Set1_dn_ffdnet_gray')
        with contextlib.redirect_stdout(io.StringIO()):
            o, _ = g['PNP_ADMM_CNC_D']('ffdnet_gray', g['mask'][k], g['noises'], **opts)
        arrays[tag] = np.asarray(o[0], np.float32)
        known[tag] = line(cap.lines)
        known[tag + '_opts'] = {kk: float(v) for kk, v in opts.items()}
        known[tag + '_sum'] = float(arrays[tag].astype(np.float64).sum())
        print(tag, known[tag], flush=True)
    # the same run cut short: a trained denoiser is locally EXPANSIVE (Lipschitz constant ~2 at the starting point, recorded below), any
    # two float32 implementations drift apart with the iteration count -- these are the counts at which 1e-5 can still be asked for
    for n_it in (2, 5, 10, 20):
        opts = dict(g['PNP_ADMM_CNC_D_opts2'], iter_num=n_it)
        with contextlib.redirect_stdout(io.StringIO()):
            o, _ = g['PNP_ADMM_CNC_D']('ffdnet_gray', g['mask'][0], g['noises'], **opts)
        tag = 'trained_cnc_d_ffdnet_gray_it%d' % n_it
        arrays[tag] = np.asarray(o[0], np.float32)
        known[tag + '_opts'] = {kk: float(v) for kk, v in opts.items()}
        known[tag + '_sum'] = float(arrays[tag].astype(np.float64).sum())
    g, _, _ = MG.run_script(S3, ['--iter_num', '1'], 'Set1_dn_drunet_gray')
    opts = dict(g['PNP_ADMM_L1_D_opts3'], iter_num=50)
    cap = MG._Capture('Set1_dn_ffdnet_gray')
    with contextlib.redirect_stdout(io.StringIO()):
        o = g['PNP_ADMM_L1_D']('ffdnet_gray', g['mask'][0], g['noises'], **opts)
    tag = 'trained_l1_d_ffdnet_gray'
    arrays[tag] = np.asarray(o[0], np.float32)
    known[tag] = line(cap.lines)
    known[tag + '_opts'] = {kk: float(v) for kk, v in opts.items()}
    known[tag + '_sum'] = float(arrays[tag].astype(np.float64).sum())
    print(tag, known[tag], flush=True)
    # the local Lipschitz constant of the trained denoiser at the loop's starting point (recorded, not asserted)
    gold_in = np.load(os.path.join(MG.GOLD, 'inputs_set1_05.npz'))
    net.eval()
    for p_ in net.parameters():
        p_.requires_grad = False
    den = D.Denoiser('ffdnet_gray', net, 15)
    mask0 = np.unpackbits(gold_in['Q_Random30_packbits'])[:65536].reshape(256, 256).astype(np.float64)
    y0 = np.fft.fft2(np.float32(gold_in['gray_u8'] / 255.)) * mask0 + gold_in['noises_c128'] * 3.0
    x0 = torch.from_numpy(np.abs(np.fft.ifft2(y0)).astype(np.float32))[None, None]
    meta.setdefault('trained', {})['lipschitz_at_x0'] = CT.lipschitz_at(lambda t: den._one(t, 0), x0, iters=12)
    tj = os.path.join(ROOT, 'gpurun_out', 'ffdnet_gray_trained.json')
    if os.path.exists(tj):
        meta['trained']['training'] = json.load(open(tj))
    meta['trained']['weights'] = 'tests/golden/ffdnet_gray_trained.npz (oracle/train_fixture_denoiser.py)'
    np.savez_compressed(npz, **arrays)
    with open(kj, 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps(meta['trained'], indent=1, sort_keys=True))


def trained_trace():
    """Round 6: TEACHER-FORCED parity at depth with a working denoiser.  A trained denoiser is locally expansive (Lipschitz ~2), so no float32
    implementation follows the reference's 50-iteration run end to end to 1e-5 -- but ONE iteration from the reference's own state can be
    held to it.  The pinned oracle loops (oracle/admm_oracle.py: pnp_admm_cnc, pnp_admm_l1) are driven here by the REFERENCE's CPU network
    -- models/network_ffdnet.FFDNet built as S6:199-213 builds it, called through the scripts' own denoising_step2 (S6:18) / denoising_step1 (S3:19) -- with
    the trained fixture weights; the run must end on the committed golden of the unmodified script (`trained_cnc_d_ffdnet_gray`, ...: that is
    what pins the recorded states to the reference), and the states (z, w) after iterations 19 / 34 / 49 together with (x, z, w) after the
    following iteration go to tests/golden/pnp_trace_set1_05.npz."""
    MG.install_shims()
    if not hasattr(np, 'int'):
        np.int = int
    d = MG.scratch_dir()
    os.chdir(d)
    os.makedirs('model_zoo')
    torch.set_num_threads(8)
    kj = os.path.join(MG.GOLD, 'pnp_known.json')
    meta = json.load(open(kj))
    for n in ('drunet_gray', 'dncnn_25', 'dncnn_15'):               # what the mains load
        net, _, _ = D.build(n)
        torch.save(D.contractive_state_dict(net, D.family(n), meta['known50']['seeds'][n], meta['gains50'][n]), os.path.join('model_zoo', n + '.pth'))
    w = np.load(os.path.join(MG.GOLD, 'ffdnet_gray_trained.npz'))
    sd = {k: torch.from_numpy(w[k]) for k in w.files}
    torch.save(sd, os.path.join('model_zoo', 'ffdnet_gray.pth'))
    from oracle import admm_oracle as O
    gold = np.load(os.path.join(MG.GOLD, 'pnp50_set1_05.npz'))
    gold_in = np.load(os.path.join(MG.GOLD, 'inputs_set1_05.npz'))
    img_L = np.float32(gold_in['gray_u8'] / 255.)                   # util.uint2single (S6:238-239)
    out, rec_meta = {}, {}
    for script, fn, tag_of, cases in ((S6, 'denoising_step2', 'trained_cnc_d_ffdnet_gray%s', (('', 0, (19, 34, 49)), ('_radial30', 1, (49,)))),
                                      (S3, 'denoising_step1', 'trained_l1_d_ffdnet_gray%s', (('', 0, (49,)),))):
        g, _, _ = MG.run_script(script, ['--iter_num', '1'], 'Set1_dn_drunet_gray')
        sys.path.insert(0, MG.REF)
        from models.network_ffdnet import FFDNet as ref_net        # the REFERENCE's class (S6:199-213)
        sys.path.remove(MG.REF)
        model = ref_net(in_nc=1, out_nc=1, nc=64, nb=15, act_mode='R')
        model.load_state_dict(sd, strict=True)
        model.eval()
        for _, v in model.named_parameters():
            v.requires_grad = False
        step = g[fn]
        cnc = script is S6
        x8 = not cnc                                               # S3:87 x8 = True (FFDNet ignores it: S3:63-65), S6:93 x8 = False

        def denoise(a, i):
            t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))[None, None]
            with torch.no_grad():
                return step('ffdnet_gray', t, x8, None, i, model, g['noises'], torch.device('cpu'), 15)[0, 0].numpy()
        for suffix, k, points in cases:
            tag = tag_of % suffix
            mask = g['mask'][k]
            y = np.fft.fft2(img_L) * mask + g['noises']
            trace = tuple(sorted(set(points) | set(p_ + 1 for p_ in points)))
            if cnc:
                o = meta['known50'][tag + '_opts']
                x, rec = O.pnp_admm_cnc(y, mask, denoise, 50, o['alpha'], o['lambda1'], o['reo'], o['b'], trace=trace)
            else:
                o = meta['known50'][tag + '_opts']
                x, rec = O.pnp_admm_l1(y, mask, denoise, 50, o['reo'], trace=trace)
            dev = float(np.linalg.norm(x.astype(np.float64) - gold[tag]) / np.linalg.norm(gold[tag]))
            print('%s: the oracle loop with the reference network ends %.3g from the unmodified script\'s golden' % (tag, dev), flush=True)
            assert dev <= 1e-6, (tag, dev)                          # pinned: the recorded states ARE the reference run's
            rec_meta[tag] = {'end_vs_golden': dev, 'points': list(points), 'opts': o}
            for p_ in points:
                out['%s_it%d_z' % (tag, p_)] = rec[p_][1]
                out['%s_it%d_w' % (tag, p_)] = rec[p_][2]
                out['%s_it%d_x' % (tag, p_ + 1)] = rec[p_ + 1][0]
                out['%s_it%d_z' % (tag, p_ + 1)] = rec[p_ + 1][1]
                out['%s_it%d_w' % (tag, p_ + 1)] = rec[p_ + 1][2]     # (w + x - z is formed from the UNCLIPPED x, z: S6:305 stands before S6:306-308)
    np.savez_compressed(os.path.join(MG.GOLD, 'pnp_trace_set1_05.npz'), **out)
    meta['trained_trace'] = rec_meta
    with open(kj, 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print({k: v.shape for k, v in out.items()})


def trained_dncnn():
    """Round 6: a TRAINED DnCNN-17 -- the x - n(x) family (models/network_dncnn.py:36-67), trained KAIR-style at sigma = 25 / 255 by
    oracle/train_fixture_denoiser.py --model dncnn_25 (tests/golden/dncnn_25_trained.npz) -- under the unmodified scripts: PNP_ADMM_CNC_DnCNN at
    the S6:571 preset (the reference loads model 1's file into both networks, S6:435) and PNP_ADMM_L1_D('dncnn_15') at the S3:341 preset, at 2 / 5
    / 10 iterations (where float32 can hold 1e-5) and, for the record and the PSNR line, the pair's own 50.  Appended to pnp50_set1_05.npz."""
    MG.install_shims()
    if not hasattr(np, 'int'):
        np.int = int
    d = MG.scratch_dir()
    os.chdir(d)
    os.makedirs('model_zoo')
    torch.set_num_threads(8)
    kj = os.path.join(MG.GOLD, 'pnp_known.json')
    meta = json.load(open(kj))
    net, _, _ = D.build('drunet_gray')                               # what the mains load besides the DnCNN files
    torch.save(D.contractive_state_dict(net, 'drunet', meta['known50']['seeds']['drunet_gray'], meta['gains50']['drunet_gray']), os.path.join('model_zoo', 'drunet_gray.pth'))
    w = np.load(os.path.join(MG.GOLD, 'dncnn_25_trained.npz'))
    net, _, _ = D.build('dncnn_25')
    sd = {k: torch.from_numpy(w[k]) for k in net.state_dict()}
    net.load_state_dict(sd, strict=True)
    for n in ('dncnn_25', 'dncnn_15'):
        torch.save(sd, os.path.join('model_zoo', n + '.pth'))
    npz = os.path.join(MG.GOLD, 'pnp50_set1_05.npz')
    arrays = dict(np.load(npz))
    known = meta['known50']

    def line(lines):
        return [l for l in lines if 'PSNR' in l and '05.png' in l][-1]
    g, _, _ = MG.run_script(S6, ['--iter_num', '1'], 'Set1_dn_drunet_gray')
    for n_it in (2, 5, 10, 50):
        opts = dict(g['PNP_ADMM_CNC_DnCNN_opts'], iter_num=n_it)
        cap = MG._Capture('Set1_dn_dncnn_25_dncnn_15')
        with contextlib.redirect_stdout(io.StringIO()):
            o, _ = g['PNP_ADMM_CNC_DnCNN']('dncnn_25', 'dncnn_15', g['mask'][0], g['noises'], **opts)
        tag = 'trained_cnc_dncnn_pair' + ('' if n_it == 50 else '_it%d' % n_it)
        arrays[tag] = np.asarray(o[0], np.float32)
        known[tag] = line(cap.lines)
        known[tag + '_opts'] = {kk: float(v) for kk, v in opts.items()}
        known[tag + '_sum'] = float(arrays[tag].astype(np.float64).sum())
        print(tag, known[tag], flush=True)
    g, _, _ = MG.run_script(S3, ['--iter_num', '1'], 'Set1_dn_drunet_gray')
    for n_it in (2, 5, 10):
        opts = dict(g['PNP_ADMM_L1_D_opts2'], iter_num=n_it)
        cap = MG._Capture('Set1_dn_dncnn_15')
        with contextlib.redirect_stdout(io.StringIO()):
            o = g['PNP_ADMM_L1_D']('dncnn_15', g['mask'][0], g['noises'], **opts)
        tag = 'trained_l1_d_dncnn_15_it%d' % n_it
        arrays[tag] = np.asarray(o[0], np.float32)
        known[tag] = line(cap.lines)
        known[tag + '_opts'] = {kk: float(v) for kk, v in opts.items()}
        known[tag + '_sum'] = float(arrays[tag].astype(np.float64).sum())
        print(tag, known[tag], flush=True)
    tj = os.path.join(ROOT, 'gpurun_out', 'dncnn_25_trained.json')
    meta.setdefault('trained_dncnn', {})
    if os.path.exists(tj):
        meta['trained_dncnn']['training'] = json.load(open(tj))
    meta['trained_dncnn']['weights'] = 'tests/golden/dncnn_25_trained.npz (oracle/train_fixture_denoiser.py --model dncnn_25)'
    np.savez_compressed(npz, **arrays)
    with open(kj, 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def MG_sigmas(nlm, iters):
    from pnp_admm_cnc_mri_amd import utils_pnp
    return utils_pnp.get_rho_sigma(max(0.255 / 255., nlm), iters, 49, nlm * 255., 1.0)[1]


def main():
    if '--extra' in sys.argv:
        return extra()
    if '--fifty' in sys.argv:
        return fifty()
    if '--trained-dncnn' in sys.argv:
        return trained_dncnn()
    if '--trained-trace' in sys.argv:
        return trained_trace()
    if '--trained' in sys.argv:
        return trained()
    MG.install_shims()
    d = MG.scratch_dir()
    os.chdir(d)
    os.makedirs('model_zoo')
    shapes = {}
    for n in NAMES:
        net, _, _ = D.build(n)
        sd = D.seeded_state_dict(net, SEEDS[n])
        torch.save(sd, os.path.join('model_zoo', n + '.pth'))
        shapes[n] = {k: list(v.shape) for k, v in sd.items()}
    torch.set_num_threads(8)
    arrays, known = {}, {'numpy': np.__version__, 'torch': torch.__version__, 'iters': ITERS, 'seeds': SEEDS,
                         'note': 'x after %d iterations, 05.png, Q_Random30, committed presets otherwise' % ITERS}

    def line(lines):
        return [l for l in lines if 'PSNR' in l and '05.png' in l][-1]

    # ---- S6: main runs drunet_gray (CNC_D) and the dncnn_25/dncnn_15 pair ----------------------
    g, lines, _ = MG.run_script(S6, ['--iter_num', str(ITERS)], 'Set1_dn_drunet_gray')
    arrays['cnc_d_drunet_gray'] = np.asarray(g['out1'], np.float32)
    arrays['cnc_dncnn_pair'] = np.asarray(g['out2'], np.float32)
    known['cnc_d_drunet_gray'] = line(lines)
    opts = {'ffdnet_gray': g['PNP_ADMM_CNC_D_opts2'], 'fdncnn_gray': g['PNP_ADMM_CNC_D_opts1']}
    for n in ('ffdnet_gray', 'fdncnn_gray'):
        cap = MG._Capture('Set1_dn_' + n)
        with contextlib.redirect_stdout(io.StringIO()):
            o, _ = g['PNP_ADMM_CNC_D'](n, g['mask'][0], g['noises'], **opts[n])
        arrays['cnc_d_' + n] = np.asarray(o[0], np.float32)
        known['cnc_d_' + n] = line(cap.lines)
        known['cnc_d_' + n + '_opts'] = {k: float(v) for k, v in opts[n].items()}
    known['cnc_d_drunet_gray_opts'] = {k: float(v) for k, v in g['PNP_ADMM_CNC_D_opts4'].items()}
    known['cnc_dncnn_pair_opts'] = {k: float(v) for k, v in g['PNP_ADMM_CNC_DnCNN_opts'].items()}

    # ---- S3: main runs drunet_gray (x8 augmentation cycles with i % 8) --------------------------
    g, lines, _ = MG.run_script(S3, ['--iter_num', str(ITERS)], 'Set1_dn_drunet_gray')
    arrays['l1_d_drunet_gray'] = np.asarray(g['out'][0], np.float32)
    o3 = {'ffdnet_gray': g['PNP_ADMM_L1_D_opts3'], 'dncnn_15': g['PNP_ADMM_L1_D_opts2'], 'fdncnn_gray': g['PNP_ADMM_L1_D_opts1']}
    known['l1_d_drunet_gray_opts'] = {k: float(v) for k, v in g['PNP_ADMM_L1_D_opts5'].items()}
    for n in ('ffdnet_gray', 'dncnn_15', 'fdncnn_gray'):
        with contextlib.redirect_stdout(io.StringIO()):
            o = g['PNP_ADMM_L1_D'](n, g['mask'][0], g['noises'], **o3[n])
        arrays['l1_d_' + n] = np.asarray(o[0], np.float32)
        known['l1_d_' + n + '_opts'] = {k: float(v) for k, v in o3[n].items()}

    for k, v in arrays.items():
        assert v.shape == (256, 256) and np.isfinite(v).all(), k
        known[k + '_sum'] = float(v.astype(np.float64).sum())
    np.savez_compressed(os.path.join(MG.GOLD, 'pnp_set1_05.npz'), **arrays)
    with open(os.path.join(MG.GOLD, 'pnp_known.json'), 'w') as f:
        json.dump({'known': known, 'state_dict_shapes': shapes}, f, indent=1, sort_keys=True)
    print(json.dumps(known, indent=1, sort_keys=True))


if __name__ == '__main__':
    main()
