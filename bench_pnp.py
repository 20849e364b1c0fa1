#!/usr/bin/env python3
"""Auxiliary benchmark (NOT the driver's headline bench.py): PnP configs of BASELINE.json on one
GPU -- config 3 (PNP_ADMM_CNC_D + FFDNet, 512 slices, Q_Radial30) and the per-GPU shard of config 4
(DRUNet, 512 slices, Q_Cartesian30) -- reporting iterations/s and the split between the HIP
x-update/glue kernels and the PyTorch-ROCm denoiser.  Seeded synthetic weights (no network).

    python bench_pnp.py --model ffdnet_gray --batch 512 --steps 10
"""
import argparse
import json
import time

import numpy as np
import torch

import pnp_admm_cnc_mri_amd as P
from pnp_admm_cnc_mri_amd import denoisers as D, solvers_pnp as SP, synthetic as S, utils_pnp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='ffdnet_gray')
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--cnn-batch', type=int, default=64)
    ap.add_argument('--mask', default=None)
    ap.add_argument('--channels-last', action='store_true')
    ap.add_argument('--cnn-dtype', default=None, choices=[None, 'bf16', 'fp16'])
    ap.add_argument('--miopen-find', action='store_true', help='torch.backends.cudnn.benchmark = True')
    args = ap.parse_args()
    fam = D.family(args.model)
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    mname = args.mask or {'ffdnet': 'Q_Radial30', 'drunet': 'Q_Cartesian30'}.get(fam, 'Q_Random30')
    mask = S.reference_masks()[mname].astype(np.uint8)
    B = args.batch
    img, noise = S.batch(0, B)
    opts = SP.PRESETS['PNP_ADMM_CNC_D'].get(fam, SP.PRESETS['PNP_ADMM_CNC_DnCNN'])
    dev = torch.device('cuda', 0)
    net, nlm, sched = D.build(args.model)
    net.load_state_dict(D.seeded_state_dict(net, 1))
    iters = args.warmup + args.steps
    sig = None
    if sched:
        sig = torch.tensor(utils_pnp.get_rho_sigma(max(0.255 / 255., nlm), iters, 49, nlm * 255., 1.0)[1])
    if args.channels_last:
        net = net.to(memory_format=torch.channels_last)
    den = D.Denoiser(args.model, net.eval(), nlm, sigmas=sig, noises=noise[0], cnn_batch=args.cnn_batch, cnn_dtype=args.cnn_dtype).to(dev)
    eng = P.Engine(256, 256, Bmax=B)
    eng.synthesize(img, noise, mask)
    eng.init_state()
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    z0, w0 = eng.get_state()
    z = torch.from_numpy(z0).to(dev).reshape(B, 1, 256, 256)
    w = torch.from_numpy(w0).to(dev).reshape(B, 1, 256, 256)
    x, s, t, zn = (torch.empty_like(z) for _ in range(4))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    t_dc = t_cnn = 0.0
    with torch.no_grad():
        for i in range(iters):
            if i == args.warmup:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                t_dc = t_cnn = 0.0
            ev[0].record()
            eng.dc_step(z, w, x, opts['reo'])
            ev[1].record()
            den(z, i, out=s)
            eng.cnc_combine(z, x, w, s, t, opts['alpha'], opts['lambda1'], opts['reo'], opts['b'])
            den(t, i, out=zn)
            ev[2].record()
            eng.dual_clamp(x, zn, w)
            ev[3].record()
            z, zn = zn, z
            torch.cuda.synchronize()
            t_dc += ev[0].elapsed_time(ev[1]) + ev[2].elapsed_time(ev[3])
            t_cnn += ev[1].elapsed_time(ev[2])
        wall = time.perf_counter() - t0
    print(json.dumps({'model': args.model, 'mask': mname, 'batch': B, 'steps': args.steps,
                      'iterations_per_s': args.steps / wall, 'slice_iterations_per_s': args.steps * B / wall,
                      'ms_per_iteration': wall / args.steps * 1e3,
                      'ms_dc_and_glue': t_dc / args.steps, 'ms_denoiser_x2': t_cnn / args.steps,
                      'x_finite': bool(torch.isfinite(x).all()), 'path': eng.path_name}))


if __name__ == '__main__':
    main()
