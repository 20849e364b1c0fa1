#!/usr/bin/env python3
"""What `python "【1】ADMM_L1.py"` / `python "【4】ADMM_CNC .py"` do (S1:171-194, S4:176-202), on
the MI355X engine: load CS_MRI/*.mat, reconstruct every image of testsets/<Set> with the committed
presets, write results/<run>/ PNGs and the reference's log lines.

    python examples/run_reference_defaults.py --root /path/to/PNP_ADMM_CNC_MRI [--solver cnc] [--mask 0]

`--root` must contain CS_MRI/ and testsets/ (the reference tree works as is; its testset directory
is `set1` while the code asks for `Set1`, so --testset defaults to whichever exists).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pnp_admm_cnc_mri_amd as P                       # noqa: E402
from pnp_admm_cnc_mri_amd import imageio               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--root', required=True)
    ap.add_argument('--solver', choices=['l1', 'cnc'], default='cnc')
    ap.add_argument('--mask', type=int, default=0, help='0 Q_Random30, 1 Q_Radial30, 2 Q_Cartesian30 (k of S4:199)')
    ap.add_argument('--testset', default=None)
    ap.add_argument('--results', default='results')
    ap.add_argument('--precision', choices=['f32', 'f64'], default='f32',
                    help="f64 = the reference's own float64 arithmetic on the device (meets 1e-5 on the committed presets)")
    # the reference's own flags (S4:21-29); unset = committed presets
    for f, t in (('alpha', float), ('iter_num', int), ('lambda1', float), ('reo', float), ('b', float)):
        ap.add_argument('--' + f, type=t, default=None)
    a = ap.parse_args()
    mask, noises = imageio.load_cs_mri(os.path.join(a.root, 'CS_MRI'))
    testsets = os.path.join(a.root, 'testsets')
    testset = a.testset or next(n for n in ('Set1', 'set1', 'set') if os.path.isdir(os.path.join(testsets, n)))
    name = 'ADMM_L1' if a.solver == 'l1' else 'ADMM_CNC'
    opts = dict(P.PRESETS[name])
    opts.update({k: v for k, v in vars(a).items() if k in opts and v is not None})
    print('------------------------------>model name = ({}) , mask = ({}) '.format(
        name, ['Q_Random30', 'Q_Radial30', 'Q_Cartesian30'][a.mask]))
    fn = P.ADMM_L1 if a.solver == 'l1' else P.ADMM_CNC
    out, info = fn(mask[a.mask], noises, testsets=testsets, testset_name=testset, results=a.results,
                   return_info=True, precision=a.precision, **opts)
    for k in ('psnr', 'ssim', 're'):
        print(k, ['%.4f' % v for v in info[k]])


if __name__ == '__main__':
    main()
