"""Solver entry points with the reference's signatures, running on the HIP engine.

    ADMM_L1(mask, noises, **opts)  -> out          ("【1】ADMM_L1.py":29-169)
    ADMM_CNC(mask, noises, **opts) -> out          ("【4】ADMM_CNC .py":31-174)

Same option names, same in-function fallback defaults (S1:35-37, S4:37-41 -- which differ from
the CLI presets S1:171 / S4:176, kept in `PRESETS`), same return value (a list pre-sized 22 whose
first entries are the x iterates, S4:50/138), same log lines (S4:155, S4:168).  What is new is
batching: all images are reconstructed at once by one fused device loop, and synthetic inputs can
bypass the file system through the extra keyword arguments

    images=   [B,H,W] uint8 (or float in [0,1]) ground-truth slices instead of testsets/<Set>
    y=        [B,H,W] complex measurements (skips the synthesis  y = fft2(img)*mask + noises)
    mask_id=  [B] index into `mask` when `mask` is a bank [K,H,W]            (build extension)
    testsets=, testset_name=, results=, save_E=, return_info=
    device=None  HIP device index; None = LOCAL_RANK when a torch.distributed process group is initialised (one process per
              GPU), else 0
    return_device=False   True: `out` is ONE torch tensor [B,H,W] on the device (float32; float64 with precision='f64') instead of the
              22-slot list of host arrays -- no device-to-host copy of the reconstructions; what sharding.solve_sharded hands
              to the final RCCL gather
    precision= 'f32' (default: float32 / complex64 device arithmetic, the throughput path) or 'f64' (the reference's own
               float64 arithmetic, S4:109: every buffer and step in double -- the setting that meets 1e-5 relative L2 on the
               committed 50-iteration CNC presets and beyond)

The PnP entry points (PNP_ADMM_L1_D, PNP_ADMM_CNC_D, PNP_ADMM_CNC_DnCNN) live in solvers_pnp.py.
"""
import logging
import os
from collections import OrderedDict

import numpy as np

from . import imageio
from .engine import Engine

# CLI presets of the reference scripts (positional order differs per script!)
PRESETS = {
    'ADMM_L1': dict(iter_num=50, lambda1=0.1, reo=0.015),                         # S1:171
    'ADMM_CNC': dict(alpha=0.45, iter_num=50, lambda1=0.5, reo=0.05, b=64),       # S4:176
}


def logger_info(logger_name, log_path):
    """File + stream logger with the reference's format (utils/utils_logger.py:25-44).  Like the
    reference, one logger per name for the whole process; unlike it, a later call with another
    `log_path` moves the file handler there instead of silently logging into the first file."""
    log = logging.getLogger(logger_name)
    formatter = logging.Formatter('%(asctime)s.%(msecs)03d : %(message)s', datefmt='%y-%m-%d %H:%M:%S')
    want = os.path.abspath(log_path)
    files = [h for h in log.handlers if isinstance(h, logging.FileHandler)]
    if not any(h.baseFilename == want for h in files):
        for h in files:
            log.removeHandler(h)
            h.close()
        fh = logging.FileHandler(log_path, mode='a')
        fh.setFormatter(formatter)
        log.addHandler(fh)
    if not any(type(h) is logging.StreamHandler for h in log.handlers):
        sh = logging.StreamHandler()
        sh.setFormatter(formatter)
        log.addHandler(sh)
    log.setLevel(logging.INFO)
    return log


def resolve_device(device=None):
    """The HIP device index an entry point runs on.  An explicit `device=` wins; the default (None) is this process's
    LOCAL_RANK when a `torch.distributed` process group is initialised -- one process per GPU: rank r of an
    8-rank job must not land on the root's card -- and 0 otherwise (the reference's single-process usage, S4:83)."""
    if isinstance(device, (int, np.integer)):
        return int(device)
    if device is not None and getattr(device, 'index', None) is not None:      # torch.device('cuda', 3)
        return int(device.index)
    if device is not None and not hasattr(device, 'index'):
        return int(device)
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return int(os.environ.get('LOCAL_RANK', 0))
    except ImportError:                                               # no torch: the plain solvers need none
        pass
    return 0


class _Job:
    """Everything around the hot loop that ADMM_L1 / ADMM_CNC / the PnP solvers share:
    inputs (S4:83-109), outputs and metrics (S4:138-172)."""

    def __init__(self, mask, noises, tag, suffix, images=None, y=None, mask_id=None, testsets='testsets',
                 testset_name='Set1', results='results', save_E=None, device=None, log=None, ssim=None,
                 psnr_fmt='{:.4f}', precision='f32'):
        # psnr_fmt: S1:150 and S3:320 print the per-image PSNR with two decimals, S4:155 / S6:332 / S6:548 with four
        self.tag, self.suffix, self.psnr_fmt = tag, suffix, psnr_fmt
        if precision not in ('f32', 'f64'):
            raise ValueError("precision must be 'f32' or 'f64'")
        self.precision = precision
        mask = np.asarray(mask)
        self.mask_bank = mask[None] if mask.ndim == 2 else mask
        self.H, self.W = self.mask_bank.shape[1:]
        self.mask_id = None if mask_id is None else np.asarray(mask_id, np.int32)
        self.names = None
        self.from_files = images is None and y is None
        if self.from_files:
            # S4:62-94: list testsets/<Set>, gray decode, modcrop(8), re-quantise
            L_path = os.path.join(testsets, testset_name)
            paths = imageio.get_image_paths(L_path)
            self.names = [os.path.basename(p) for p in paths]
            imgs = [imageio.modcrop(imageio.imread_gray(p), 8) for p in paths]
            for p, im in zip(paths, imgs):
                if im.shape != (self.H, self.W):
                    raise ValueError('%s is %s, mask is %dx%d' % (p, im.shape, self.H, self.W))
            images = np.stack(imgs)
            self.E_path = os.path.join(results, testset_name + '_dn_' + tag)
            self.save_E = True if save_E is None else save_E
        else:
            self.E_path = os.path.join(results, testset_name + '_dn_' + tag)
            self.save_E = bool(save_E)
        self.gt_u8 = None
        self.img_L = None
        if images is not None:
            images = np.asarray(images)
            if images.ndim == 2:
                images = images[None]
            if images.dtype != np.uint8:
                images = imageio.single2uint(np.asarray(images, np.float32))
            self.gt_u8 = np.ascontiguousarray(images)
            self.img_L = imageio.requantise(self.gt_u8)                          # S4:91-94
        self.y = None if y is None else np.asarray(y)
        if self.y is not None and self.y.ndim == 2:
            self.y = self.y[None]
        self.B = len(self.gt_u8) if self.gt_u8 is not None else len(self.y)
        self.noises = None if noises is None else np.asarray(noises)
        self.device = resolve_device(device)
        self.ssim = True if ssim is None else ssim
        self.log = log
        if self.log is None and (self.from_files or self.save_E):
            os.makedirs(self.E_path, exist_ok=True)
            name = testset_name + '_dn_' + tag
            self.log = logger_info(name, os.path.join(self.E_path, name + '.log'))
            self.log.info(os.path.join(testsets, testset_name))
        self.testset_name = testset_name

    def open_engine(self, stream=None):
        """stream: HIP stream handle every engine call is ordered on (PnP: torch's current stream),
        set BEFORE the first kernel so upload / synthesis / init and the loop share one queue."""
        eng = Engine(self.H, self.W, Bmax=self.B, device=self.device, precision=self.precision)
        if stream is not None:
            eng.set_stream(stream)
        if self.y is not None:
            eng.upload(self.y, self.mask_bank, self.mask_id)
        else:
            eng.synthesize(self.img_L, self.noises, self.mask_bank, self.mask_id)        # S4:102
        eng.init_state()                                                                  # S4:103-109
        return eng

    def finish(self, eng, x, x_dev=None, extra=''):
        """S4:138-172: out list, optional PNGs, PSNR/SSIM/RE log lines, averages.  Metrics are
        device reductions on x_dev (None = the ctx's x; PnP passes the uint8-quantised x, S6:314)."""
        A = np.zeros((self.H, self.W), dtype='uint8')
        psnr1 = [0] * max(22, self.B)
        if isinstance(x, np.ndarray):
            out = [A] * max(22, self.B)
            for n in range(self.B):
                out[n] = x[n].astype(np.float64)
        else:
            out = x                                                  # return_device=True: the device tensor [B,H,W] itself
            if self.save_E:
                x = x.reshape(self.B, self.H, self.W).cpu().numpy()
        info = OrderedDict(psnr=[], ssim=[], re=[])
        if self.gt_u8 is not None:
            psnr, re = eng.metrics(x_dev, self.gt_u8)                                     # device reductions
            info['psnr'], info['re'] = list(map(float, psnr)), list(map(float, re))
            if self.ssim:
                info['ssim'] = list(map(float, eng.ssim(x_dev, self.gt_u8)))                  # device, double
            for n in range(self.B):
                psnr1[n] = info['psnr'][n]
                if self.log is not None and self.names is not None:
                    self.log.info(('{:s} - PSNR: ' + self.psnr_fmt + ' dB; SSIM: {:.4f} ; RE: {:.4f}.').format(
                        self.names[n], info['psnr'][n], info['ssim'][n] if self.ssim else float('nan'), info['re'][n]))
            if self.log is not None:
                ave = lambda v: sum(v) / len(v) if v else float('nan')
                self.log.info('------> testset_name: ({}), {}Average PSNR:({:.3f})dB, Average ssim : ({:.3f}), '
                              'Average re : ({:.3f}) )'.format(self.testset_name, extra, ave(info['psnr']),
                                                               ave(info['ssim']), ave(info['re'])))
        if self.save_E:
            os.makedirs(self.E_path, exist_ok=True)
            for n in range(self.B):
                stem = os.path.splitext(self.names[n])[0] if self.names else '%04d' % n
                imageio.imsave_gray(x[n] * 255, os.path.join(self.E_path, stem + self.suffix + '.png'))
        return out, psnr1, info


def _device_x(eng, job):
    """the ctx's x copied device-to-device into a torch tensor that outlives the engine"""
    import torch
    xt = torch.empty((job.B, job.H, job.W), dtype=torch.float64 if job.precision == 'f64' else torch.float32,
                     device=torch.device('cuda', job.device))
    eng.x(out=xt)
    eng.sync()
    return xt


def ADMM_L1(mask, noises, images=None, y=None, mask_id=None, testsets='testsets', testset_name='Set1',
            results='results', save_E=None, device=None, return_info=False, precision='f32', return_device=False, **ADMM_L1_opts):
    """ADMM with L1 prox on the MI355X engine.  Reference: "【1】ADMM_L1.py":29-169."""
    iter_num = ADMM_L1_opts.get('iter_num', 20)          # S1:35
    lambda1 = ADMM_L1_opts.get('lambda1', 0.04)          # S1:36
    reo = ADMM_L1_opts.get('reo', 0.04)                  # S1:37
    job = _Job(mask, noises, 'ADMM_L1', '_PDG L1', images, y, mask_id, testsets, testset_name, results, save_E, device,
               psnr_fmt='{:.2f}', precision=precision)   # S1:150
    with job.open_engine() as eng:
        eng.admm_l1(iter_num, lambda1, reo)              # S1:111-126, all slices, on device
        x = _device_x(eng, job) if return_device else eng.x()      # iter_num = 0: the initial x = |ifft2(y)| (S4:103, 138)
        out, _, info = job.finish(eng, x)
    return (out, info) if return_info else out


def ADMM_CNC(mask, noises, images=None, y=None, mask_id=None, testsets='testsets', testset_name='Set1',
             results='results', save_E=None, device=None, return_info=False, precision='f32', return_device=False, **ADMM_CNC_opts):
    """ADMM with the convex-non-convex z-step.  Reference: "【4】ADMM_CNC .py":31-174."""
    iter_num = ADMM_CNC_opts.get('iter_num', 4)          # S4:37
    alpha = ADMM_CNC_opts.get('alpha', 0.4)              # S4:38
    lambda1 = ADMM_CNC_opts.get('lambda1', 0.04)         # S4:39
    reo = ADMM_CNC_opts.get('reo', 2.75)                 # S4:40  (reo is 1/beta of the paper)
    b = ADMM_CNC_opts.get('b', 1)                        # S4:41  (b is b^2 of the paper)
    job = _Job(mask, noises, 'ADMM_CNC', '_ADMM CNC', images, y, mask_id, testsets, testset_name, results, save_E, device,
               precision=precision)
    with job.open_engine() as eng:
        eng.admm_cnc(iter_num, alpha, lambda1, reo, b)   # S4:115-132
        x = _device_x(eng, job) if return_device else eng.x()      # iter_num = 0: the initial x = |ifft2(y)| (S4:103, 138)
        out, _, info = job.finish(eng, x)
    return (out, info) if return_info else out
