#!/usr/bin/env python3
"""The f16x3 conv3x3 (kernels_conv_f16x3.hip) on DRUNet's residual-block shapes at 64 slices per call -- C -> C channels at
[64, 64, 256, 256], [64, 128, 128, 128], [64, 256, 64, 64], [64, 512, 32, 32] (equal arithmetic per layer) -- against MIOpen
(channels_last, find mode on): distance from the float64 result and time per layer.
usage (GPU box): python3 profiles/experiments/probe_conv_wide.py [n]"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pnp_admm_cnc_mri_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) >= 2 else 64
L = _lib.lib()
dev = torch.device('cuda', 0)
torch.manual_seed(0)
torch.backends.cudnn.benchmark = True
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for ch, hw in ((64, 256), (128, 128), (256, 64), (512, 32)):
    x = torch.randn(n, ch, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(ch, ch, 3, 3, device=dev) * (2.0 / (9 * ch)) ** 0.5
    sk = torch.randn(n, ch, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    xn, skn = x.permute(0, 2, 3, 1), sk.permute(0, 2, 3, 1)
    assert xn.is_contiguous()
    pk = torch.empty(9 * ch * ch, device=dev)
    _lib.check(L.pnp_conv3x3_pack_f16x3(stream, p(w), p(pk), ch))
    y = torch.empty_like(xn)
    run = lambda: _lib.check(L.pnp_conv3x3_nhwc_f16x3(stream, p(xn), p(pk), None, p(skn), p(y), n, ch, hw, hw, 0))
    run()
    m = min(n, 4)
    ref = F.conv2d(x[:m].double(), w.double(), padding=1) + sk[:m].double()
    wcl = w.contiguous(memory_format=torch.channels_last)
    mi = lambda: F.conv2d(x, wcl, padding=1).add_(sk)
    e_h, e_m = rel(y[:m].permute(0, 3, 1, 2), ref), rel(mi()[:m], ref)
    t_h, t_m = timeit(run), timeit(mi)
    t_c = timeit(lambda: F.conv2d(x, wcl, padding=1))
    flop = 2.0 * n * hw * hw * ch * ch * 9
    print('C %3d [%d, %d, %d, %d]  f16x3 %.3f ms %.1f TF (%.2e from f64) | miopen conv + add %.3f ms %.1f TF (%.2e), conv alone %.3f ms %.1f TF'
          % (ch, n, ch, hw, hw, t_h, flop / t_h / 1e9, e_h, t_m, flop / t_m / 1e9, e_m, t_c, flop / t_c / 1e9))
