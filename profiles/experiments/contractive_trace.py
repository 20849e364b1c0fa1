"""per-iteration distance between the float32-CNN and float64-CNN oracle loops (contractive fixture weights): growth pattern"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pnp_admm_cnc_mri_amd import denoisers as D, solvers_pnp as SP, utils_pnp
from oracle import admm_oracle as O, contractive as CT
torch.set_num_threads(8)
gold = np.load(os.path.join(ROOT, 'tests/golden/inputs_set1_05.npz'))
gray = gold['gray_u8']; noises = gold['noises_c128'] * 3.0
mask = np.unpackbits(gold['Q_Random30_packbits'])[:65536].reshape(256, 256).astype(np.float64)
y = O.synthesize(np.float32(gray / 255.), mask, noises)
name, loop = sys.argv[1], sys.argv[2]
fam = D.family(name)
net, nlm, sched = D.build(name)
net.load_state_dict(D.contractive_state_dict(net, fam, 2000, CT.conv_operator_norms(net, 2000)))
net.eval()
iters = 50
sig = torch.tensor(utils_pnp.get_rho_sigma(max(0.255 / 255., nlm), iters, 49, nlm * 255., 1.0)[1]) if sched else None
tr = tuple(range(1, iters + 1))
rec = {}
for dt in (torch.float32, torch.float64):
    netd = net.to(dt)
    den = D.Denoiser(name, netd, nlm, sigmas=sig, noises=noises)
    den.model = netd
    if den.noise_map is not None:
        den.noise_map = den.noise_map.to(dt)
    def dn(a, i, den=den, dt=dt):
        with torch.no_grad():
            return den._one(torch.from_numpy(np.ascontiguousarray(a)).to(dt)[None, None], i)[0, 0].float().numpy()
    if loop == 'cnc':
        p = SP.PRESETS['PNP_ADMM_CNC_D'].get(fam, SP.PRESETS['PNP_ADMM_CNC_DnCNN'])
        _, rec[dt] = O.pnp_admm_cnc(y, mask, dn, iters, p['alpha'], p['lambda1'], p['reo'], p['b'], trace=tr)
    else:
        _, rec[dt] = O.pnp_admm_l1(y, mask, dn, iters, SP.PRESETS['PNP_ADMM_L1_D'][fam]['reo'], trace=tr)
for i in tr:
    a, b = rec[torch.float32][i], rec[torch.float64][i]
    d = [np.linalg.norm(a[k] - b[k]) / max(np.linalg.norm(b[k]), 1e-30) for k in range(3)]
    sat = [float((b[k] <= 0).mean()) for k in range(3)] + [float((b[k] >= 1).mean()) for k in range(3)]
    print(i, 'x %.2e z %.2e w %.2e' % tuple(d), 'frac at 0: x %.3f z %.3f w %.3f   at 1: x %.3f z %.3f w %.3f' % tuple(sat))
for i in (30, 40, 45, 50):
    a, b = rec[torch.float32][i], rec[torch.float64][i]
    d = np.abs(a[0] - b[0])
    idx = np.unravel_index(np.argmax(d), d.shape)
    print(i, 'x: max |d| %.2e at %s, pixels with |d| > 1e-5: %d, > 1e-6: %d; values there f32 x %.6f z %.6f w %.6f | f64 x %.6f z %.6f w %.6f'
          % (d.max(), idx, (d > 1e-5).sum(), (d > 1e-6).sum(), a[0][idx], a[1][idx], a[2][idx], b[0][idx], b[1][idx], b[2][idx]))
