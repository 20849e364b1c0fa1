"""sensitivity of the 50-iteration oracle loops to a 1e-7 relative perturbation of every denoiser output (float32 CNN on the CPU), for
variants of the contractive fixture weights:  contractive_sens.py family loop [a_last] [bias_scale] [eps_body] [eps_head]"""
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
if len(sys.argv) > 3: D._CONTRACTIVE_TAIL[fam] = float(sys.argv[3])
bias_scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
if len(sys.argv) > 5: D._CONTRACTIVE['body'] = (1.0, float(sys.argv[5]))
if len(sys.argv) > 6: D._CONTRACTIVE['head'] = (1.0, float(sys.argv[6]))
net, nlm, sched = D.build(name)
sd = D.contractive_state_dict(net, fam, 2000, CT.conv_operator_norms(net, 2000))
for k in sd:
    if k.endswith('bias'): sd[k] = sd[k] * bias_scale
net.load_state_dict(sd); net.eval()
iters = 50
sig = torch.tensor(utils_pnp.get_rho_sigma(max(0.255 / 255., nlm), iters, 49, nlm * 255., 1.0)[1]) if sched else None
den = D.Denoiser(name, net, nlm, sigmas=sig, noises=noises, x8=(loop == 'l1' and fam in ('drunet', 'ffdnet')))
tr = (10, 20, 30, 40, 50)
rec = {}
for pert in (0, 1):
    rng = np.random.default_rng(5)
    def dn(a, i):
        with torch.no_grad():
            o = den._one(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))[None, None], i)[0, 0].numpy()
        return o * (1 + 1e-7 * rng.standard_normal(o.shape).astype(np.float32)) if pert else o
    if loop == 'cnc':
        p = SP.PRESETS['PNP_ADMM_CNC_D'].get(fam, SP.PRESETS['PNP_ADMM_CNC_DnCNN'])
        x, rec[pert] = O.pnp_admm_cnc(y, mask, dn, iters, p['alpha'], p['lambda1'], p['reo'], p['b'], trace=tr)
    else:
        x, rec[pert] = O.pnp_admm_l1(y, mask, dn, iters, SP.PRESETS['PNP_ADMM_L1_D'][fam]['reo'], trace=tr)
print(name, loop, sys.argv[3:], 'PSNR %.3f' % O.calculate_psnr(np.round(x * 255.), gray),
      ' '.join('it%d %.1e' % (i, np.linalg.norm(rec[0][i][0] - rec[1][i][0]) / np.linalg.norm(rec[0][i][0])) for i in tr), flush=True)
