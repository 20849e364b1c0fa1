#!/usr/bin/env python3
"""Full-length PnP runs (the reference's presets: 50 iterations, S6:569-577) on two synthetic slices with the CNN forward on the three
backends -- PyTorch / MIOpen, float32 MFMA ('hip'), split-half f16 MFMA ('hip_f16x3') -- same seeded weights: rel-L2 of the final x against
the PyTorch backend's and the PSNR the solver reports.   usage (GPU box): python3 profiles/experiments/full_length_pnp_backends.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pnp_admm_cnc_mri_amd import denoisers as D, synthetic as S, solvers_pnp as SP
mask = S.reference_masks()['Q_Radial30'].astype(np.uint8)
img, noise = S.batch(300, 2)
for name in ('ffdnet_gray', 'fdncnn_gray', 'ircnn_gray', 'drunet_gray'):
    res = {}
    opts = dict(SP.PRESETS['PNP_ADMM_CNC_D'][D.family(name)])
    for backend in ('torch', 'hip', 'hip_f16x3'):
        net, _, _ = D.build(name)
        net.load_state_dict(D.seeded_state_dict(net, 1))
        out, psnr = SP.PNP_ADMM_CNC_D(name, mask, noise[0], images=img, model=net, results='/tmp/res_full', cnn_backend=backend,
                                      miopen_find=False, **opts)
        res[backend] = (np.stack([out[b] for b in range(2)]), psnr)
    for b in ('hip', 'hip_f16x3'):
        e = np.linalg.norm(res[b][0] - res['torch'][0]) / np.linalg.norm(res['torch'][0])
        print('%-12s %d iterations  %-10s rel-L2 vs the PyTorch backend %.3e   PSNR %s (PyTorch backend %s)'
              % (name, opts['iter_num'], b, e, [round(float(v), 4) for v in res[b][1]], [round(float(v), 4) for v in res['torch'][1]]))
