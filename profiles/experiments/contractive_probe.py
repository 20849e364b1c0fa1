"""Probe of the contractive fixture weights on the CPU: operator-norm gains, local Lipschitz constant of D, and how far a float32-CNN
loop and a float64-CNN loop of the oracle end from each other after the presets' 50 iterations (the sensitivity every pair of float32
implementations shares).  python profiles/experiments/contractive_probe.py [family ...]"""
import sys, time, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pnp_admm_cnc_mri_amd import denoisers as D, solvers_pnp as SP, utils_pnp
from oracle import admm_oracle as O, contractive as CT

torch.set_num_threads(8)
gold = np.load(os.path.join(ROOT, 'tests/golden/inputs_set1_05.npz'))
gray = gold['gray_u8']
noises = gold['noises_c128'] * 3.0
mask = np.unpackbits(gold['Q_Random30_packbits'])[:65536].reshape(256, 256).astype(np.float64)
img = np.float32(gray / 255.)
y = O.synthesize(img, mask, noises)

names = sys.argv[1:] or ['ffdnet_gray', 'dncnn_15', 'fdncnn_gray', 'ircnn_gray', 'drunet_gray']
for name in names:
    fam = D.family(name)
    net, nlm, sched = D.build(name)
    t0 = time.time()
    gains = CT.conv_operator_norms(net, 2000)
    t1 = time.time()
    sd = D.contractive_state_dict(net, fam, 2000, gains)
    net.load_state_dict(sd)
    net.eval()
    iters = 50
    sig = None
    if sched:
        sig = torch.tensor(utils_pnp.get_rho_sigma(max(0.255 / 255., nlm), iters, 49, nlm * 255., 1.0)[1])
    outs = {}
    for dt in (torch.float32, torch.float64):
        netd = net.to(dt)
        den = D.Denoiser(name, netd, nlm, sigmas=sig, noises=noises)
        den.model = netd
        if den.noise_map is not None:
            den.noise_map = den.noise_map.to(dt)

        def dn(a, i, den=den, dt=dt):
            with torch.no_grad():
                t = torch.from_numpy(np.ascontiguousarray(a)).to(dt)[None, None]
                return den._one(t, i)[0, 0].float().numpy()
        if dt == torch.float32:
            x0 = torch.from_numpy(np.abs(np.fft.ifft2(y)).astype(np.float32))[None, None]
            lip = CT.lipschitz_at(lambda t: den._one(t, 0), x0, iters=12)
            with torch.no_grad():
                d0 = den._one(x0, 0)
            print(name, 'gains %.1fs' % (t1 - t0), 'Lip(D) at x0 ~ %.4f' % lip, '|D(x0)|/|x0| %.4f' % float(d0.norm() / x0.norm()),
                  '|D(x0)-x0|/|x0| %.4f' % float((d0 - x0).norm() / x0.norm()), flush=True)
        p = SP.PRESETS['PNP_ADMM_CNC_D'].get(fam, SP.PRESETS['PNP_ADMM_CNC_DnCNN'])
        t2 = time.time()
        xc = O.pnp_admm_cnc(y, mask, dn, iters, p['alpha'], p['lambda1'], p['reo'], p['b'])
        pl = SP.PRESETS['PNP_ADMM_L1_D'][fam]
        xl = O.pnp_admm_l1(y, mask, dn, iters, pl['reo'])
        outs[dt] = (xc, xl)
        print('   ', dt, 'loops %.1fs' % (time.time() - t2), 'PSNR cnc %.3f l1 %.3f' % (O.calculate_psnr(np.round(xc * 255.), gray), O.calculate_psnr(np.round(xl * 255.), gray)), flush=True)
    for k, tag in ((0, 'cnc'), (1, 'l1')):
        a, b = outs[torch.float32][k], outs[torch.float64][k]
        print('    f32-CNN vs f64-CNN loop, %s, 50 it: rel-L2 %.3e' % (tag, np.linalg.norm(a - b) / np.linalg.norm(b)), flush=True)
