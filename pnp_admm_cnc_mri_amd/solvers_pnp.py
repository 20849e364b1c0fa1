"""Plug-and-play solver entry points with the reference's signatures:

    PNP_ADMM_L1_D(model_name, mask, noises, **opts)                 -> out            (S3:77-337)
    PNP_ADMM_CNC_D(model_name, mask, noises, **opts)                -> (out, psnr1)   (S6:79-351)
    PNP_ADMM_CNC_DnCNN(model_name1, model_name2, mask, noises, **o) -> (out, psnr1)   (S6:372-567)

(S3 = "【3】PNP_ADMM_L1_D  .py", S6 = "【6】PNP_ADMM_CNC_D .py".)  The x-update runs in the HIP
engine (pnp_dc_step on torch's current stream), the denoiser is a PyTorch-ROCm module, the
pointwise glue (S6:301, 305-308) runs in HIP kernels on the tensors' device pointers.  State never
leaves the device; what the reference's host<->device marshalling does to the numbers is kept:
float32 state, |z| and |w| (no-ops once clamped), clamp(0,1) of x, z and w after every iteration.

Extra keyword arguments (all optional): images=, y=, mask_id=, testsets=, testset_name=, results=,
save_E=, device=, return_info=  as in solvers.py, plus
    model_zoo='model_zoo'   directory of KAIR .pth files (S6:107-109)
    model= / model2=        an nn.Module (or state_dict) instead of a file
    cnn_batch=None          slices per CNN forward (activation memory).  None: 64 -- 256 (of 256 x 256; the same pixel count for larger slices) for
                            the plain stacks on the HIP backends, whose layers are short launches at 64 (denoisers.Denoiser)
    cnn_dtype=None          None = float32 (parity); 'bf16'/'fp16' = autocast throughput mode, off parity
    miopen_find='auto'      'auto': torch.backends.cudnn.benchmark is switched on around the CNN forward passes of conv batches
                            of >= 16 images and restored afterwards (a process-global flag: INTEGRATION.md section 4);
                            False / True: leave the caller's setting alone / force a find for every forward
    cnn_backend='auto'      'auto' (default since round 6): 'hip_f16x3' when the device is gfx950, every convolution of the network is one
                            libpnpmri.so takes and the weights lie inside the half range, else 'torch' -- one log line says which and why
                            (denoisers.auto_backend); 'torch': the whole CNN forward in PyTorch-ROCm / MIOpen (north star); 'hip': the 64 -> 64 conv3x3 + ReLU
                            body layers of DnCNN / FDnCNN / FFDNet on libpnpmri.so's fp32 matrix-core kernel (exact f32
                            arithmetic, ~1e-6 from MIOpen's results; first and last layer on its direct kernels);
                            'hip_f16x3': the same with the 64 -> 64 layers in split-half arithmetic on the f16 matrix cores
                            (float32 operands as two halves, exact products, float32 accumulation: float32-level results,
                            2.3 x the float32 kernel's rate; operands must lie within the half range)
    return_device=False     True: `out` is one torch tensor [B,H,W] on the device instead of the 22-slot list of host arrays (solvers.py)
    state0=None, iter_start=0   (PNP_ADMM_CNC_D, PNP_ADMM_L1_D) resume the loop: state0 = (z, w) arrays [B,H,W] as they stand AFTER iteration
                            iter_start, the loop then runs iterations iter_start .. iter_num - 1 (sigma schedule, bank switch and x8 mode of
                            those indices).  With return_info=True the final (z, w) come back as info['z'], info['w'].  What the
                            teacher-forced parity tests use: one iteration from the reference's own state (tests/test_gpu_pnp.py)
    cnn_graph=False         True: a denoiser forward of at most cnn_batch slices is captured once per shape into a HIP graph and replayed
                            (the reference's one-slice calls: a forward is a train of short launches; FFDNet 0.50 -> see DESIGN.md 4.8)
"""
import os

import numpy as np

from . import denoisers as D
from . import utils_pnp as pnp
from .solvers import _Job, resolve_device

PRESETS = {
    # PNP_ADMM_CNC_D(alpha, iter, lambda1, reo, b), S6:569-577
    'PNP_ADMM_CNC_D': {
        'fdncnn': dict(alpha=0.9, iter_num=50, lambda1=0.2, reo=0.45, b=0.3),
        'ffdnet': dict(alpha=0.9, iter_num=50, lambda1=1.35, reo=0.45, b=0.3),
        'ircnn': dict(alpha=0.5, iter_num=50, lambda1=1.3, reo=0.45, b=2),
        'drunet': dict(alpha=1, iter_num=50, lambda1=0.8, reo=0.8, b=0.45),
    },
    'PNP_ADMM_CNC_DnCNN': dict(alpha=1.2, iter_num=50, lambda1=4, reo=0.45, b=0.3),     # S6:571
    # PNP_ADMM_L1_D(iter, reo), S3:339-347
    'PNP_ADMM_L1_D': {
        'fdncnn': dict(iter_num=50, reo=0.25), 'dncnn': dict(iter_num=50, reo=0.15),
        'ffdnet': dict(iter_num=50, reo=0.25), 'ircnn': dict(iter_num=50, reo=0.145),
        'drunet': dict(iter_num=50, reo=0.26),
    },
}


def _load_model(model_name, model, model_zoo, iter_num, noises, x8, cnn_batch, device, cnn_dtype=None, miopen_find='auto',
                cnn_backend='auto', cnn_graph=False):
    """The model-zoo switch of S6:129-217: build by name, load weights, eval, no grad, to device."""
    import torch
    net, nlm, scheduled = D.build(model_name)
    bank = None
    if model is None:
        path = os.path.join(model_zoo, model_name + '.pth')
        if not os.path.exists(path):
            raise FileNotFoundError('%s not found (the reference ships no weights, model_zoo/README.md); pass model= '
                                    'or put KAIR weights there' % path)
        model = torch.load(path, map_location='cpu')
    if isinstance(model, torch.nn.Module):
        net = model
    elif D.family(model_name) == 'ircnn' and all(k.isdigit() for k in model.keys()):
        bank = model                                   # dict of 25 state_dicts keyed by str(index), S6:196
        net.load_state_dict(bank['0'] if '0' in bank else next(iter(bank.values())), strict=True)
    else:
        net.load_state_dict(model, strict=True)
    net.eval()
    for _, v in net.named_parameters():
        v.requires_grad = False
    sigmas = None
    if scheduled:                                      # S6:172-174, 191-193
        _, s = pnp.get_rho_sigma(sigma=max(0.255 / 255., nlm), iter_num=iter_num, modelSigma1=49,
                                 modelSigma2=nlm * 255., w=1.0)
        sigmas = torch.tensor(s)
    return D.Denoiser(model_name, net, nlm, sigmas=sigmas, noises=noises, x8=x8, bank=bank, cnn_batch=cnn_batch,
                      cnn_dtype=cnn_dtype, miopen_find=miopen_find, backend=cnn_backend, graph=cnn_graph).to(device)


def _device_state(torch, eng, B, H, W, dev):
    """z0 = |ifft2(y)|, w0 = 0 (S6:253-256) copied device-to-device into torch tensors; x starts as
    z0 too (S6:252-253), which is what a zero-iteration call returns."""
    z = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
    w = torch.empty_like(z)
    eng.get_state(z, w)
    x = z.clone()
    return x, z, w


def _resume(torch, z, w, state0, dev):
    """state0 = (z, w) [B,H,W] host arrays -> the loop's device tensors (float32, as S6:277-285 hands them to the network)"""
    if state0 is None:
        return
    z0, w0 = state0
    z.copy_(torch.from_numpy(np.ascontiguousarray(z0, dtype=np.float32)).reshape(z.shape).to(dev))
    w.copy_(torch.from_numpy(np.ascontiguousarray(w0, dtype=np.float32)).reshape(w.shape).to(dev))


def _finish_pnp(torch, job, eng, x, extra, return_device=False):
    """S6:314-351: img_E = uint8(round(x*255)); metrics on the quantised image.  return_device: `out` is the device tensor [B,H,W]
    (solvers.py), no host copy of the reconstructions."""
    xq = torch.round(x * 255.0) / 255.0
    xs = x.reshape(job.B, job.H, job.W)
    out, psnr1, info = job.finish(eng, xs.clone() if return_device else xs.cpu().numpy(), x_dev=xq.contiguous(), extra=extra)
    return out, psnr1, info


def PNP_ADMM_CNC_D(model_name, mask, noises, images=None, y=None, mask_id=None, testsets='testsets',
                   testset_name='Set1', results='results', save_E=None, device=None, return_info=False,
                   model_zoo='model_zoo', model=None, cnn_batch=None, cnn_dtype=None, miopen_find='auto', cnn_backend='auto', cnn_graph=False, return_device=False,
                   state0=None, iter_start=0, **PNP_ADMM_CNC_D_opts):
    """CNC ADMM with a CNN denoiser in place of both soft-thresholds.  Reference: S6:79-351."""
    import torch
    alpha = PNP_ADMM_CNC_D_opts.get('alpha', 0.4)          # S6:85-89
    iter_num = PNP_ADMM_CNC_D_opts.get('iter_num', 46)
    lambda1 = PNP_ADMM_CNC_D_opts.get('lambda1', 2.75)
    reo = PNP_ADMM_CNC_D_opts.get('reo', 1)
    b = PNP_ADMM_CNC_D_opts.get('b', 1)
    device = resolve_device(device)                                    # None: LOCAL_RANK under a process group, else 0
    dev = torch.device('cuda', device)
    job = _Job(mask, noises, model_name, 'PNP_ADMM_CNC_D', images, y, mask_id, testsets, testset_name, results,
               save_E, device)
    den = _load_model(model_name, model, model_zoo, iter_num, noises, False, cnn_batch, dev, cnn_dtype, miopen_find, cnn_backend, cnn_graph)   # x8 = False, S6:93
    with torch.cuda.device(dev), torch.no_grad(), job.open_engine(torch.cuda.current_stream(dev).cuda_stream) as eng:
        B, H, W = job.B, job.H, job.W
        x, z, w = _device_state(torch, eng, B, H, W, dev)
        _resume(torch, z, w, state0, dev)
        s = torch.empty_like(z)
        t = torch.empty_like(z)
        z_new = torch.empty_like(z)
        for i in range(iter_start, iter_num):                                 # S6:262
            eng.dc_step(z, w, x, reo)                                         # S6:266-271
            den.select_bank(i)                                                # S6:289-298
            den(z, i, out=s)                                                  # S6:300
            eng.cnc_combine(z, x, w, s, t, alpha, lambda1, reo, b)            # S6:301
            den(t, i, out=z_new)                                              # S6:302
            eng.dual_clamp(x, z_new, w)                                       # S6:305-308
            z, z_new = z_new, z
        torch.cuda.current_stream(dev).synchronize()
        out, psnr1, info = _finish_pnp(torch, job, eng, x, 'alpha: ({:.3f}), '.format(alpha), return_device)
        if return_info:
            info['z'], info['w'] = z.reshape(B, H, W).cpu().numpy(), w.reshape(B, H, W).cpu().numpy()
    return (out, psnr1, info) if return_info else (out, psnr1)


def PNP_ADMM_CNC_DnCNN(model_name1, model_name2, mask, noises, images=None, y=None, mask_id=None,
                       testsets='testsets', testset_name='Set1', results='results', save_E=None, device=None,
                       return_info=False, model_zoo='model_zoo', model=None, model2=None, cnn_batch=None,
                       cnn_dtype=None, faithful_model2_path=True, miopen_find='auto', cnn_backend='auto', cnn_graph=False, return_device=False, **opts):
    """Two DnCNN-17 nets: s = D1(z), z = D2(t).  Reference: S6:372-567.
    `faithful_model2_path`: the reference loads model_path1 into BOTH nets (S6:435) although it logs
    path 2; True reproduces that, False loads model_name2's own weights."""
    import torch
    alpha = opts.get('alpha', 0.4)
    iter_num = opts.get('iter_num', 46)
    lambda1 = opts.get('lambda1', 2.75)
    reo = opts.get('reo', 1)
    b = opts.get('b', 1)
    device = resolve_device(device)                                    # None: LOCAL_RANK under a process group, else 0
    dev = torch.device('cuda', device)
    job = _Job(mask, noises, model_name1 + '_' + model_name2, 'PNP_ADMM_CNC_DnCNN', images, y, mask_id, testsets,
               testset_name, results, save_E, device)
    den1 = _load_model(model_name1, model, model_zoo, iter_num, noises, False, cnn_batch, dev, cnn_dtype, miopen_find, cnn_backend, cnn_graph)
    if faithful_model2_path and model2 is None:
        den2 = _load_model(model_name1, model, model_zoo, iter_num, noises, False, cnn_batch, dev, cnn_dtype, miopen_find, cnn_backend, cnn_graph)
    else:
        den2 = _load_model(model_name2, model2, model_zoo, iter_num, noises, False, cnn_batch, dev, cnn_dtype, miopen_find, cnn_backend, cnn_graph)
    with torch.cuda.device(dev), torch.no_grad(), job.open_engine(torch.cuda.current_stream(dev).cuda_stream) as eng:
        B, H, W = job.B, job.H, job.W
        x, z, w = _device_state(torch, eng, B, H, W, dev)
        s, t, z_new = torch.empty_like(z), torch.empty_like(z), torch.empty_like(z)
        for i in range(iter_num):                                             # S6:491
            eng.dc_step(z, w, x, reo)                                         # S6:495-500
            den1(z, i, out=s)                                                 # S6:517
            eng.cnc_combine(z, x, w, s, t, alpha, lambda1, reo, b)            # S6:518
            den2(t, i, out=z_new)                                             # S6:519
            eng.dual_clamp(x, z_new, w)                                       # S6:522-525
            z, z_new = z_new, z
        torch.cuda.current_stream(dev).synchronize()
        out, psnr1, info = _finish_pnp(torch, job, eng, x, 'alpha: ({:.3f}), '.format(alpha), return_device)
    return (out, psnr1, info) if return_info else (out, psnr1)


def PNP_ADMM_L1_D(model_name, mask, noises, images=None, y=None, mask_id=None, testsets='testsets',
                  testset_name='Set1', results='results', save_E=None, device=None, return_info=False,
                  model_zoo='model_zoo', model=None, cnn_batch=None, cnn_dtype=None, miopen_find='auto', cnn_backend='auto', cnn_graph=False, return_device=False,
                  state0=None, iter_start=0, **PNP_ADMM_L1_D_opts):
    """L1-ADMM with the CNN as the prox: z = D(x + w).  Reference: S3:77-337."""
    import torch
    iter_num = PNP_ADMM_L1_D_opts.get('iter_num', 20)      # S3:83-84
    reo = PNP_ADMM_L1_D_opts.get('reo', 0.04)
    device = resolve_device(device)                                    # None: LOCAL_RANK under a process group, else 0
    dev = torch.device('cuda', device)
    fam = D.family(model_name)
    x8 = fam in ('drunet', 'ffdnet')                       # x8 = True (S3:87) survives only there (S3:130,142,181)
    job = _Job(mask, noises, model_name, '_' + model_name + '_PNP_ADMM_L1_D', images, y, mask_id, testsets,
               testset_name, results, save_E, device, psnr_fmt='{:.2f}')            # file name S3:308, PSNR format S3:320
    den = _load_model(model_name, model, model_zoo, iter_num, noises, x8, cnn_batch, dev, cnn_dtype, miopen_find, cnn_backend, cnn_graph)
    with torch.cuda.device(dev), torch.no_grad(), job.open_engine(torch.cuda.current_stream(dev).cuda_stream) as eng:
        B, H, W = job.B, job.H, job.W
        x, z, w = _device_state(torch, eng, B, H, W, dev)
        _resume(torch, z, w, state0, dev)
        t = torch.empty_like(z)
        for i in range(iter_start, iter_num):                                 # S3:255
            eng.dc_step(z, w, x, reo)                                         # S3:259-264
            den.select_bank(i)
            eng.add(x, w, t)                                                  # x + w
            den(t, i, out=z)                                                  # S3:290
            eng.dual_clamp(x, z, w)                                           # S3:293-296
        torch.cuda.current_stream(dev).synchronize()
        out, _, info = _finish_pnp(torch, job, eng, x, '', return_device)
        if return_info:
            info['z'], info['w'] = z.reshape(B, H, W).cpu().numpy(), w.reshape(B, H, W).cpu().numpy()
    return (out, info) if return_info else out
