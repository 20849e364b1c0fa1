"""MIOpen at DRUNet's one-slice convolution shapes (PNP_ADMM_CNC_D, backend 'torch', 10 iterations = 20 forwards), by what the process did first.
Measured (round 5, one MI355X box per line group):
  A  torch.backends.cudnn.deterministic = True, immediate mode, 3 iterations:   9.5 s      (naive kernels: ~1 s per forward)
     then deterministic = False + find, 10 iterations:                          19.7 s, again 19.8 s   <- the process keeps the naive kernels
  B  fresh process, deterministic = False, find:                                5.6 s (the search), again 0.2 s
  C  fresh process, deterministic = False, immediate mode only:                 3.6 s (compilation), again 0.2 s
So the product's default (no deterministic flag; find for batches >= 16, immediate mode below) is fast either way; what is slow is the deterministic
flag of tests/test_gpu_pnp.py's module fixture, and it stays slow for the rest of that process -- hence tests/pnp50_runner.py in a fresh process.
usage (GPU box): python3 profiles/experiments/miopen_immediate_vs_find.py A|B|C"""
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
elif mode == 'B':
    torch.backends.cudnn.deterministic = False
    print('fresh det=False find=True, 10 it: %.1f s' % run(10, cnn_backend='torch', miopen_find=True))
    print('again: %.1f s' % run(10, cnn_backend='torch', miopen_find=True))
    print('det=False find=False, 10 it: %.1f s' % run(10, cnn_backend='torch', miopen_find=False))
if mode == 'C':      # fresh process, immediate mode only (what miopen_find='auto' did for batches below 16 until round 5)
    torch.backends.cudnn.deterministic = False
    print('fresh det=False find=False (immediate mode), 10 it: %.1f s' % run(10, cnn_backend='torch', miopen_find=False))
    print('again: %.1f s' % run(10, cnn_backend='torch', miopen_find=False))
