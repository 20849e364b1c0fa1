import os, sys, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import pnp_admm_cnc_mri_amd as P
from pnp_admm_cnc_mri_amd import synthetic as S
from oracle import admm_oracle as O
mask = S.reference_masks()['Q_Random30'].astype(np.uint8)
def rel(a,b): return float(np.linalg.norm(a.astype(np.float64)-b)/np.linalg.norm(b))
ids = [227, 3, 99, 355, 483]
img = np.stack([S.phantom(b) for b in ids]); noise = np.stack([S.kspace_noise(b) for b in ids])
for mode, fast in (('1', 1), ('0', 1), ('0', 0)):
    os.environ['PNP_SLICE'] = mode
    with P.Engine(256, 256, Bmax=len(ids)) as eng:
        eng.set_fast_path(fast)
        eng.synthesize(img, noise, mask[None], np.zeros(len(ids), np.int32))
        y = eng.download_y()
        out = {}
        for it in (5, 10, 15, 20, 25):
            eng.init_state(); eng.admm_cnc(it, 0.45, 0.5, 0.05, 64); out[it] = eng.x()
        path = eng.path_name
    for k, b in enumerate(ids):
        y128 = y[k].astype(np.complex128)
        print(path, 'slice', b, ' '.join('it%d %.2e (f32 numpy %.2e)' % (it, rel(out[it][k], O.admm_cnc(y128, mask, it)), rel(O.admm_cnc_f32(y128, mask, it), O.admm_cnc(y128, mask, it))) for it in (5, 15, 25)))
