"""Run ON THE GPU BOX: the slice-resident path (PNP_SLICE=1) against the two-launch path (PNP_SLICE=0) for 20 batch sizes from 1 to 300
(odd sizes, sizes around the 64-slice selection threshold and around one round of 256), 6 CNC + 5 L1 iterations, three masks:
x, z, w must agree to float round-off (max-abs <= 2e-5 on values in [0, 1])."""
import os, sys, subprocess, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
code = r'''
import sys, os, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import pnp_admm_cnc_mri_amd as P
from pnp_admm_cnc_mri_amd import synthetic as S
m = S.reference_masks(); masks = np.stack([m['Q_Random30'], m['Q_Cartesian30'], m['Q_Radial30']]).astype(np.uint8)
out = {}
for B in [1, 2, 3, 5, 7, 8, 9, 15, 16, 17, 31, 33, 63, 64, 65, 100, 255, 256, 257, 300]:
    img, noise = S.batch(0, B)
    with P.Engine(256, 256, Bmax=B) as eng:
        eng.synthesize(img, noise, masks, np.arange(B) % 3); eng.init_state()
        eng.admm_cnc(6, 0.45, 0.5, 0.05, 64); x = eng.x(); z, w = eng.get_state()
        eng.init_state(); eng.admm_l1(5, 0.1, 0.015); xl = eng.x()
        out['%d_path' % B] = np.array([eng.path_name == 'slice'])
        out['%d_x' % B], out['%d_z' % B], out['%d_w' % B], out['%d_xl' % B] = x, z, w, xl
np.savez(sys.argv[1], **out)
'''
res = {}
for mode in ('1', '0'):
    path = '/tmp/sweep_%s.npz' % mode
    subprocess.check_call([sys.executable, '-c', code, path], env=dict(os.environ, PNP_SLICE=mode))
    res[mode] = np.load(path)
worst = 0
for k in res['1'].files:
    if k.endswith('_path'):
        continue
    a, b = res['1'][k].astype(np.float64), res['0'][k].astype(np.float64)
    e = np.abs(a - b).max()                       # images and state live in [0, 1]: absolute error (w is small, its relative error says little)
    worst = max(worst, e)
    if e > 2e-5:
        print('MISMATCH', k, e)
print('batch sizes checked: slice path vs two-launch path, worst max-abs difference over x, z, w, x(L1): %.3e' % worst)
