"""how far does the DnCNN-pair loop (S6:571 preset, contractive fixture weights, CPU float32 CNN) end from itself when ONLY the x-update
runs in complex64 / float32 instead of NumPy's float64 -- the part every float32 GPU engine shares"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from pnp_admm_cnc_mri_amd import denoisers as D, solvers_pnp as SP
from oracle import admm_oracle as O
from conftest import weights50
torch.set_num_threads(8)
gold = np.load(os.path.join(ROOT, 'tests/golden/inputs_set1_05.npz'))
gray = gold['gray_u8']; noises = gold['noises_c128'] * 3.0
mask = np.unpackbits(gold['Q_Random30_packbits'])[:65536].reshape(256, 256).astype(np.float64)
y = O.synthesize(np.float32(gray / 255.), mask, noises)
name = sys.argv[1] if len(sys.argv) > 1 else 'dncnn_25'
fam = D.family(name)
net, nlm, _ = D.build(name)
if len(sys.argv) > 2:
    import json
    meta = json.load(open(os.path.join(ROOT, 'tests/golden/pnp_known.json')))
    D._CONTRACTIVE_TAIL[fam] = float(sys.argv[2])
    net.load_state_dict(D.contractive_state_dict(net, fam, meta['known50']['seeds'][name], meta['gains50'][name]))
else:
    net.load_state_dict(weights50(name))
net.eval()
den = D.Denoiser(name, net, nlm, noises=noises)
def dn(a, i):
    with torch.no_grad():
        return den._one(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))[None, None], i)[0, 0].numpy()
p = SP.PRESETS['PNP_ADMM_CNC_D'].get(fam, SP.PRESETS['PNP_ADMM_CNC_DnCNN'])
tr = (1, 2, 5, 10, 20, 30, 40, 50)
x64, r64 = O.pnp_admm_cnc(y, mask, dn, 50, p['alpha'], p['lambda1'], p['reo'], p['b'], trace=tr)
orig = O.dc_step
y32 = y.astype(np.complex64)
def dc32(z, w, yy, m, reo):
    index = np.nonzero(m)
    xf = np.fft.fft2((np.float32(z) - np.float32(w)).astype(np.float32))
    assert xf.dtype == np.complex64
    La2 = np.float32(1.0 / 2.0 / reo)
    xf[index] = (La2 * xf[index] + y32[index]) / (np.float32(1.0) + La2)
    return np.absolute(np.real(np.fft.ifft2(xf)))
O.dc_step = dc32
x32, r32 = O.pnp_admm_cnc(y, mask, dn, 50, p['alpha'], p['lambda1'], p['reo'], p['b'], trace=tr)
O.dc_step = orig
print(name, ' '.join('it%d %.2e' % (i, np.linalg.norm(r32[i][0] - r64[i][0]) / np.linalg.norm(r64[i][0])) for i in tr))
ref = np.load(os.path.join(ROOT, 'tests/golden/pnp50_set1_05.npz'))
if name == 'dncnn_25' and len(sys.argv) <= 2:
    print('vs golden: f64 %.2e  f32-xupdate %.2e' % (np.linalg.norm(x64 - ref['cnc_dncnn_pair']) / np.linalg.norm(ref['cnc_dncnn_pair']),
                                                       np.linalg.norm(x32 - ref['cnc_dncnn_pair']) / np.linalg.norm(ref['cnc_dncnn_pair'])))
