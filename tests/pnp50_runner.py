#!/usr/bin/env python3
"""The measured rel-L2 of every 50-iteration PnP golden (tests/golden/pnp50_set1_05.npz) on each CNN backend -- the numbers behind the 1e-5
assertions of tests/test_gpu_pnp.py.  Also the FRESH PROCESS in which that file runs DRUNet on the MIOpen-backed backends: once MIOpen has
served a convolution shape in immediate mode under torch's deterministic flag (the rest of that test module needs it), a later find in the same
process no longer helps -- a one-slice DRUNet forward stays at ~1 s (naive kernels) where a fresh process with find enabled needs 20 ms
(profiles/experiments/miopen_immediate_vs_find.py).  TEST INFRASTRUCTURE (it checks against tests/golden and uses the oracle's PSNR).
usage (GPU box): python3 tests/pnp50_runner.py [backends...] [--tags tag ...]"""
import json, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from conftest import weights50, weights_trained, GOLD
from pnp_admm_cnc_mri_amd import solvers_pnp as S
from oracle import admm_oracle as O
args = sys.argv[1:]
only = None
if '--tags' in args:                                   # pnp50_runner.py [backends...] --tags tag [tag ...]
    k = args.index('--tags')
    only, args = set(args[k + 1:]), args[:k]
backends = args or ['torch', 'hip', 'hip_f16x3']
d = np.load(os.path.join(GOLD, 'inputs_set1_05.npz'))
masks = {k[:-9]: np.unpackbits(d[k])[:65536].reshape(256, 256).astype(np.float64) for k in d.files if k.endswith('_packbits')}
gray, noises = d['gray_u8'], d['noises_c128'] * 3.0
known = json.load(open(os.path.join(GOLD, 'pnp_known.json')))['known50']
gold = np.load(os.path.join(GOLD, 'pnp50_set1_05.npz'))
rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
psnr = lambda x: O.calculate_psnr(np.round(x.astype(np.float64) * 255), gray)
tmp = tempfile.mkdtemp()
torch.backends.cudnn.deterministic = False
results = []
for tag in sorted(gold.files):
    if only is not None and tag not in only:
        continue
    for be in backends:
        t0 = time.time()
        kw = dict(images=gray[None], results=tmp, cnn_backend=be, miopen_find=True)
        if tag == 'cnc_dncnn_pair':
            o = {k: (int(v) if k == 'iter_num' else v) for k, v in known[tag + '_opts'].items()}
            out, _ = S.PNP_ADMM_CNC_DnCNN('dncnn_25', 'dncnn_15', masks['Q_Random30'], noises, model=weights50('dncnn_25'), **kw, **o)
        elif tag.startswith('trained_'):
            name = 'ffdnet_gray'
            o = {k: (int(v) if k == 'iter_num' else v) for k, v in known[tag + '_opts'].items()}
            if 'l1_d' in tag:
                out = S.PNP_ADMM_L1_D(name, masks['Q_Random30'], noises, model=weights_trained(), **kw, **o)
            else:
                out, _ = S.PNP_ADMM_CNC_D(name, masks['Q_Radial30' if tag.endswith('radial30') else 'Q_Random30'], noises, model=weights_trained(), **kw, **o)
        elif tag.startswith('cnc_d_'):
            parts = tag[6:].split('_'); name = '_'.join(parts[:2])
            m = masks[{'radial30': 'Q_Radial30', 'cartesian30': 'Q_Cartesian30'}.get(parts[-1], 'Q_Random30')]
            o = {k: (int(v) if k == 'iter_num' else v) for k, v in known[tag + '_opts'].items()}
            out, _ = S.PNP_ADMM_CNC_D(name, m, noises, model=weights50(name), **kw, **o)
        else:
            name = tag[5:]
            o = {k: (int(v) if k == 'iter_num' else v) for k, v in known[tag + '_opts'].items()}
            out = S.PNP_ADMM_L1_D(name, masks['Q_Random30'], noises, model=weights50(name), **kw, **o)
        print('%-32s %-10s rel-L2 %.2e   PSNR %.4f (golden %.4f)   %.0f s' % (tag, be, rel(out[0], gold[tag]), psnr(out[0]), psnr(gold[tag]), time.time() - t0), flush=True)
        results.append({'tag': tag, 'backend': be, 'rel_l2': rel(out[0], gold[tag]), 'psnr': psnr(out[0]), 'psnr_golden': psnr(gold[tag]),
                        'log_line': known.get(tag), 'seconds': time.time() - t0})
print('JSON ' + json.dumps(results))
