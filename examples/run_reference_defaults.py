#!/usr/bin/env python3
"""What `python "【1】ADMM_L1.py"` / `python "【4】ADMM_CNC .py"` do (S1:171-194, S4:176-202) -- and, with a model name, what
`python "【3】PNP_ADMM_L1_D  .py"` / `python "【6】PNP_ADMM_CNC_D .py"` do (S3:339-380, S6:569-620) -- on the MI355X engine: load
CS_MRI/*.mat, reconstruct every image of testsets/<Set> with the committed presets, write results/<run>/ PNGs and the reference's
log lines.

    python examples/run_reference_defaults.py --root /path/to/PNP_ADMM_CNC_MRI [--solver cnc] [--mask 0]
    python examples/run_reference_defaults.py --root ... --solver pnp_cnc --model drunet_gray [--model-zoo model_zoo] [--cnn-backend hip_f16x3]
    python examples/run_reference_defaults.py --root ... --solver pnp_cnc --model dncnn_25 --model2 dncnn_15      (the DnCNN pair, S6:571)

The PnP solvers need KAIR weights `<model-zoo>/<model>.pth` (the reference ships none: model_zoo/README.md).

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
    ap.add_argument('--solver', choices=['l1', 'cnc', 'pnp_l1', 'pnp_cnc'], default='cnc')
    ap.add_argument('--model', default=None, help='pnp_*: fdncnn_gray | dncnn_15 | ffdnet_gray | ircnn_gray | drunet_gray (name[m] of S3:372 / S6:605)')
    ap.add_argument('--model2', default=None, help='pnp_cnc: the second DnCNN of PNP_ADMM_CNC_DnCNN (S6:617)')
    ap.add_argument('--model-zoo', default=None, help='directory of the .pth files (default: <root>/model_zoo)')
    ap.add_argument('--cnn-backend', default='auto', choices=['auto', 'torch', 'hip', 'hip_f16x3'])
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
    if a.solver.startswith('pnp'):
        return pnp_main(a, mask, noises, testsets, testset)
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


def pnp_main(a, mask, noises, testsets, testset):
    from pnp_admm_cnc_mri_amd import denoisers as D, solvers_pnp as SP
    if not a.model:
        raise SystemExit('--solver pnp_* needs --model')
    fam = D.family(a.model)
    zoo = a.model_zoo or os.path.join(a.root, 'model_zoo')
    kw = dict(testsets=testsets, testset_name=testset, results=a.results, return_info=True, model_zoo=zoo, cnn_backend=a.cnn_backend)
    over = {k: v for k, v in vars(a).items() if k in ('alpha', 'iter_num', 'lambda1', 'reo', 'b') and v is not None}
    print('------------------------------>model name = ({}) , mask = ({}) '.format(a.model, ['Q_Random30', 'Q_Radial30', 'Q_Cartesian30'][a.mask]))
    if a.solver == 'pnp_l1':
        opts = dict(SP.PRESETS['PNP_ADMM_L1_D'][fam])                                  # S3:339-347
        opts.update({k: v for k, v in over.items() if k in opts})
        out, info = SP.PNP_ADMM_L1_D(a.model, mask[a.mask], noises, **kw, **opts)
    elif a.model2:
        opts = dict(SP.PRESETS['PNP_ADMM_CNC_DnCNN'])                                  # S6:571
        opts.update(over)
        out, _, info = SP.PNP_ADMM_CNC_DnCNN(a.model, a.model2, mask[a.mask], noises, **kw, **opts)
    else:
        opts = dict(SP.PRESETS['PNP_ADMM_CNC_D'][fam])                                 # S6:569-577
        opts.update(over)
        out, _, info = SP.PNP_ADMM_CNC_D(a.model, mask[a.mask], noises, **kw, **opts)
    for k in ('psnr', 'ssim', 're'):
        print(k, ['%.4f' % v for v in info[k]])


if __name__ == '__main__':
    main()
