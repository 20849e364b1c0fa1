#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: a TRAINED FFDNet-gray for the parity fixtures.

The reference ships no weights (model_zoo/README.md: download links only) and the build container has no network, so every PnP
fixture so far used seeded synthetic weights.  This script trains the reference's FFDNet architecture (models/network_ffdnet.py:31-73,
here pnp_admm_cnc_mri_amd.denoisers.FFDNet with KAIR's state_dict keys) as KAIR trains it -- Gaussian noise of a random level sigma in
[0, 75] / 255 on clean patches, the level handed to the network as its noise map, L1 loss, Adam -- on seeded synthetic images (ellipse
phantoms x band-limited textures), for a few minutes on one GPU.  What comes out is no state-of-the-art denoiser; it is a network whose
weights were shaped by training: structured filters, the activation statistics of a working denoiser, a map that really removes noise
(the held-out PSNR gains are recorded).  That is what the f16x3 backend's operand range and the 50-iteration PnP parity had not met yet.

Training is not bit-reproducible, so the RESULT is the fixture: tests/golden/ffdnet_gray_trained.npz (float32 arrays under the
state_dict keys, 1.8 MB) + the 'trained' entry of tests/golden/pnp_known.json (seed, steps, losses, held-out PSNR).  The goldens of the
unmodified reference scripts with these weights come from oracle/make_golden_pnp.py --trained (CPU, build container).

usage (GPU box): python3 oracle/train_fixture_denoiser.py [--steps 6000] [--out gpurun_out/ffdnet_gray_trained.npz]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pnp_admm_cnc_mri_amd import denoisers as D, synthetic as S      # noqa: E402


def images(n, seed):
    """n seeded 256 x 256 training images in [0, 1]: an ellipse phantom modulated by a band-limited random texture, plus a few sharp
    line and point structures (edges and detail for the network to keep)"""
    rng = np.random.default_rng(seed)
    fy, fx = np.fft.fftfreq(256)[:, None], np.fft.fftfreq(256)[None, :]
    out = np.empty((n, 256, 256), np.float32)
    for k in range(n):
        ph = S.phantom(100000 + seed * 1000 + k).astype(np.float64)
        cut = rng.uniform(0.03, 0.2)
        spec = (rng.standard_normal((256, 256)) + 1j * rng.standard_normal((256, 256))) * np.exp(-(fy ** 2 + fx ** 2) / (2 * cut ** 2))
        tex = np.real(np.fft.ifft2(spec))
        tex = tex / (np.abs(tex).max() + 1e-12)
        img = ph * (0.75 + 0.25 * tex) + 0.08 * np.maximum(tex, 0) * (ph > 0.02)
        for _ in range(rng.integers(2, 6)):                      # thin bright / dark lines
            a, c = rng.uniform(0, np.pi), rng.uniform(40, 216, 2)
            yy, xx = np.mgrid[0:256, 0:256]
            d = np.abs((yy - c[0]) * np.cos(a) - (xx - c[1]) * np.sin(a))
            img += rng.uniform(-0.15, 0.25) * np.exp(-(d / rng.uniform(0.6, 1.5)) ** 2) * (ph > 0.02)
        out[k] = np.clip(img, 0, 1)
    return out


def psnr(a, b):
    return float(10 * torch.log10(1.0 / torch.mean((a - b) ** 2)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=6000)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--patch', type=int, default=96)
    ap.add_argument('--seed', type=int, default=20261005)
    ap.add_argument('--model', default='ffdnet_gray', choices=['ffdnet_gray', 'drunet_gray', 'dncnn_25'],
                    help='drunet_gray: for profiles/experiments/drunet_trained_check.py only (32 M parameters: nothing is committed); dncnn_25 (round 6): '
                         'DnCNN-17, the x - n(x) family, trained KAIR-style at ONE noise level (25 / 255) -> tests/golden/dncnn_25_trained.npz')
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    if a.out is None:
        a.out = os.path.join(ROOT, 'gpurun_out', a.model + '_trained.npz')
    dev = torch.device('cuda', 0)
    torch.manual_seed(a.seed)
    t0 = time.time()
    train = torch.from_numpy(images(192, 1)).to(dev)
    held = torch.from_numpy(images(16, 2)).to(dev)
    print('images: %.1f s' % (time.time() - t0), flush=True)
    net, _, _ = D.build(a.model)
    net = net.to(dev).train()
    drunet, dncnn = a.model == 'drunet_gray', a.model == 'dncnn_25'

    def run(noisy, sigma):
        """FFDNet takes the level as its second argument; DRUNet as a second input channel (S6:36-38); DnCNN none (one level per model)"""
        if drunet:
            return net(torch.cat((noisy, sigma.expand(-1, 1, noisy.shape[2], noisy.shape[3])), 1))
        if dncnn:
            return net(noisy)
        return net(noisy, sigma)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4 if a.model == 'drunet_gray' else 1e-3)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, a.steps, eta_min=2e-5)
    g = torch.Generator(device=dev).manual_seed(a.seed)
    torch.backends.cudnn.benchmark = True
    losses = []
    for step in range(a.steps):
        idx = torch.randint(0, train.shape[0], (a.batch,), device=dev, generator=g)
        oy = int(torch.randint(0, 256 - a.patch + 1, (1,), generator=g, device=dev))
        ox = int(torch.randint(0, 256 - a.patch + 1, (1,), generator=g, device=dev))
        clean = train[idx, oy:oy + a.patch, ox:ox + a.patch][:, None]
        if step & 1:
            clean = clean.flip(3)
        if step & 2:
            clean = clean.transpose(2, 3)
        sigma = torch.rand((a.batch, 1, 1, 1), device=dev, generator=g) * (75.0 / 255.0)
        if dncnn:
            sigma = torch.full_like(sigma, 25.0 / 255.0)
        noisy = clean + sigma * torch.randn(clean.shape, device=dev, generator=g)
        loss = torch.nn.functional.l1_loss(run(noisy, sigma), clean)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        sched.step()
        losses.append(float(loss))
        if step % 500 == 0 or step + 1 == a.steps:
            print('step %5d  L1 %.5f  lr %.2e  %.0f s' % (step, np.mean(losses[-100:]), sched.get_last_lr()[0], time.time() - t0), flush=True)
    net.eval()
    rec = {'seed': a.seed, 'steps': a.steps, 'batch': a.batch, 'patch': a.patch, 'final_l1': float(np.mean(losses[-200:])), 'held_out_psnr': {}}
    with torch.no_grad():
        for s in ((25,) if dncnn else (15, 25, 50)):
            sig = torch.full((held.shape[0], 1, 1, 1), s / 255.0, device=dev)
            noisy = held[:, None] + sig * torch.randn(held[:, None].shape, device=dev, generator=g)
            den = run(noisy, sig)
            rec['held_out_psnr'][str(s)] = {'noisy': psnr(noisy, held[:, None]), 'denoised': psnr(den, held[:, None])}
            print('sigma %d: noisy %.2f dB -> denoised %.2f dB' % (s, rec['held_out_psnr'][str(s)]['noisy'], rec['held_out_psnr'][str(s)]['denoised']), flush=True)
    sd = {k: v.detach().float().cpu().numpy() for k, v in net.state_dict().items()}
    rec['max_abs_weight'] = float(max(np.abs(v).max() for v in sd.values()))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    np.savez_compressed(a.out, **sd)
    with open(a.out[:-4] + '.json', 'w') as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    print(json.dumps(rec, sort_keys=True))


if __name__ == '__main__':
    main()
