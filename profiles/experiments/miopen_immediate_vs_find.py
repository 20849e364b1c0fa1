import sys, os, time, tempfile
ROOT = os.getcwd(); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, json
from conftest import weights50, GOLD
from pnp_admm_cnc_mri_amd import solvers_pnp as S
d = np.load(os.path.join(GOLD, 'inputs_set1_05.npz'))
mask = np.unpackbits(d['Q_Random30_packbits'])[:65536].reshape(256, 256).astype(np.float64)
gray, noises = d['gray_u8'], d['noises_c128'] * 3.0
tmp = tempfile.mkdtemp()
w = weights50('drunet_gray')
o = dict(alpha=1, lambda1=0.8, reo=0.8, b=0.45)
def run(iters, **kw):
    t0 = time.time()
    S.PNP_ADMM_CNC_D('drunet_gray', mask, noises, images=gray[None], model=w, results=tmp, iter_num=iters, **o, **kw)
    return time.time() - t0
mode = sys.argv[1]
if mode == 'A':      # as the test module: deterministic first, immediate mode; then find
    torch.backends.cudnn.benchmark = False; torch.backends.cudnn.deterministic = True
    print('det=True immediate, 3 it: %.1f s' % run(3, cnn_backend='torch'))
    torch.backends.cudnn.deterministic = False
    print('then det=False find=True, 10 it: %.1f s' % run(10, cnn_backend='torch', miopen_find=True))
    print('again: %.1f s' % run(10, cnn_backend='torch', miopen_find=True))
else:
    torch.backends.cudnn.deterministic = False
    print('fresh det=False find=True, 10 it: %.1f s' % run(10, cnn_backend='torch', miopen_find=True))
    print('again: %.1f s' % run(10, cnn_backend='torch', miopen_find=True))
    print('det=False find=False, 10 it: %.1f s' % run(10, cnn_backend='torch', miopen_find=False))
