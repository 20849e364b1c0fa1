#!/usr/bin/env python3
"""DRUNet with TRAINED weights on the split-half backend (no fixture: 32 M parameters are not committed; train in the same gpurun call:
python3 oracle/train_fixture_denoiser.py --model drunet_gray --steps 2500 --batch 32).  Checks, against the PyTorch / MIOpen backend with the
same weights: one forward on noisy seeded images (rel-L2), every activation inside the half range (PNP_CONV_CHECK_RANGE=1), and
PNP_ADMM_CNC_D on the reference's image at 5 / 10 / 20 iterations (rel-L2 between the backends, PSNR of each).
usage (GPU box): python3 profiles/experiments/drunet_trained_check.py gpurun_out/drunet_gray_trained.npz"""
import os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pnp_admm_cnc_mri_amd import denoisers as D, solvers_pnp as SP, synthetic as S
from oracle import admm_oracle as O
w = np.load(sys.argv[1])
sd = {k: torch.from_numpy(w[k]) for k in w.files}
print('max |w| %.3f' % max(float(v.abs().max()) for v in sd.values()))
g = np.load(os.path.join(ROOT, 'tests/golden/inputs_set1_05.npz'))
gray, noises = g['gray_u8'], g['noises_c128'] * 3.0
mask = np.unpackbits(g['Q_Cartesian30_packbits'])[:65536].reshape(256, 256).astype(np.float64)
rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.linalg.norm(np.asarray(b, np.float64)))
torch.backends.cudnn.deterministic = False
dens = {}
for be in ('torch', 'hip_f16x3'):
    net, nlm, _ = D.build('drunet_gray')
    net.load_state_dict(sd)
    dens[be] = D.Denoiser('drunet_gray', net.eval(), nlm, sigmas=torch.tensor([25 / 255.]), backend=be, miopen_find=True).to('cuda')
clean = torch.from_numpy(np.stack([S.phantom(900 + k) for k in range(8)]))[:, None].cuda()
noisy = clean + (25 / 255.) * torch.randn(clean.shape, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1))
a = dens['torch'](noisy, 0)
os.environ['PNP_CONV_CHECK_RANGE'] = '1'
b = dens['hip_f16x3'](noisy, 0)
os.environ.pop('PNP_CONV_CHECK_RANGE')
psnr = lambda x: float(10 * torch.log10(1 / ((x - clean) ** 2).mean()))
print('forward, 8 x 256^2, sigma 25: noisy %.2f dB, torch %.2f dB, hip_f16x3 %.2f dB; hip_f16x3 vs torch rel-L2 %.2e; range check silent' % (psnr(noisy), psnr(a), psnr(b), rel(b.cpu(), a.cpu())))
tmp = tempfile.mkdtemp()
for n_it in (5, 10, 20):
    res = {}
    for be in ('torch', 'hip_f16x3'):
        out, _ = SP.PNP_ADMM_CNC_D('drunet_gray', mask, noises, images=gray[None], model=sd, results=tmp, cnn_backend=be, miopen_find=True,
                                   **dict(SP.PRESETS['PNP_ADMM_CNC_D']['drunet'], iter_num=n_it))
        res[be] = out[0]
    p = {be: O.calculate_psnr(np.round(res[be] * 255), gray) for be in res}
    print('PNP_ADMM_CNC_D, Q_Cartesian30, %2d iterations: torch %.4f dB, hip_f16x3 %.4f dB, rel-L2 between them %.2e' % (n_it, p['torch'], p['hip_f16x3'], rel(res['hip_f16x3'], res['torch'])))
