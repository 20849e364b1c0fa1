import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pnp_admm_cnc_mri_amd as P
from pnp_admm_cnc_mri_amd import synthetic as S
from oracle import admm_oracle as O
B = int(os.environ.get('DBG_B', '5'))
mk = S.reference_masks()
masks = np.stack([mk['Q_Random30'], mk['Q_Radial30'], mk['Q_Cartesian30']]).astype(np.uint8)
mid = (np.arange(B) % 3).astype(np.int32)
ys = np.stack([O.synthetic_problem(b, masks[mid[b]])[1] for b in range(B)]).astype(np.complex64)
def run(slice_on, iters, solver='cnc'):
    os.environ['PNP_SLICE'] = slice_on
    with P.Engine(256, 256, Bmax=B) as eng:
        eng.upload(ys, masks, mid)
        eng.init_state()
        if solver == 'cnc': eng.admm_cnc(iters, 0.45, 0.5, 0.05, 64)
        else: eng.admm_l1(iters, 0.1, 0.015)
        return eng.x(), eng.get_state()
(x1, (z1, w1)) = run('0', 1)
(x2, (z2, w2)) = run('0', 2)
os.environ['PNP_SLICE'] = '0'
with P.Engine(256, 256, Bmax=B) as eng:
    eng.upload(ys, masks, mid); eng.init_state(); z0, w0 = eng.get_state()
# what a stale read of (z0, w0) in the last phase would give: prox(x2, z0, w0)
zs0 = np.empty_like(z0); ws0 = np.empty_like(w0)
for b in range(B):
    zz, ww = O.cnc_step(x2[b].astype(np.float64), z0[b].astype(np.float64), w0[b].astype(np.float64), 0.45, 0.5, 0.05, 64)
    zs0[b], ws0[b] = zz, ww
for k in range(3):
    (xs, (zs, ws)) = run('1', 2)
    bad = np.argwhere(np.abs(zs - z2) > 1e-4)
    print('run', k, 'nbad', len(bad))
    if len(bad):
        idx = tuple(bad.T)
        print('  equals z after 1 iteration (last store lost):', np.mean(np.abs(zs[idx] - z1[idx]) < 1e-5))
        print('  equals prox(x2, z0, w0) (stale loads):        ', np.mean(np.abs(zs[idx] - zs0[idx]) < 1e-5), np.mean(np.abs(ws[idx] - ws0[idx]) < 1e-5))
        print('  equals z0:', np.mean(np.abs(zs[idx] - z0[idx]) < 1e-5))
        print('  rows', np.unique(bad[:, 1]), 'lanes', np.unique(bad[:, 2] // 4))
        b0, r0, c0 = bad[0]
        c0 = (c0 // 4) * 4
        np.set_printoptions(precision=5, linewidth=200)
        for nm, arr in (('zs', zs), ('z2', z2), ('z1', z1), ('z0', z0), ('ws', ws), ('w2', w2), ('w1', w1), ('x2', x2), ('xs', xs), ('z2 row+1', None), ('zs0', zs0)):
            if arr is None:
                print('   z2[row+1]', z2[b0, r0 + 1, c0 - 4:c0 + 8]); continue
            print('  ', nm, arr[b0, r0, c0 - 4:c0 + 8])
        # does the wrong z equal prox(x2, z1, w_something)?  try w = 0 and w = w0
        for nm, (zz, ww) in (('(z1,w1)', (z1, w1)), ('(z1,w0)', (z1, w0)), ('(z0,w1)', (z0, w1)), ('(z1,0)', (z1, 0 * w1))):
            zt, wt = O.cnc_step(x2[b0].astype(np.float64), zz[b0].astype(np.float64), np.asarray(ww[b0], np.float64), 0.45, 0.5, 0.05, 64)
            print('   prox(x2,%s) z' % nm, zt[r0, c0 - 4:c0 + 8], 'match', np.mean(np.abs(zt[tuple(bad[bad[:,0]==b0][:,1:].T)] - zs[b0][tuple(bad[bad[:,0]==b0][:,1:].T)]) < 1e-5))
