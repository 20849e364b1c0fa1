#!/usr/bin/env python3
"""The f16x3 64 -> 64 conv3x3 layer at FFDNet's body shape [n, 64, 128, 128] by activation format (pnp_conv3x3_nhwc_f16x3_fmt) and by
data: float32 -> float32 (round 4's layer), float32 -> split, split -> split (what 11 of FFDNet's 13 body layers run since round 5),
split -> float32; random activations and all-zero ones (the power probe: same instruction stream, no toggling operands).
usage (GPU box): python3 profiles/experiments/probe_conv_fmt.py [n=64] [only_fmt]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pnp_admm_cnc_mri_amd import _lib, denoisers as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
only = int(sys.argv[2]) if len(sys.argv) > 2 else None
H = W = 128
L = _lib.lib()
torch.manual_seed(0)
dev = torch.device('cuda', 0)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
w = torch.randn(64, 64, 3, 3, device=dev) * (2.0 / 576) ** 0.5
b = torch.randn(64, device=dev) * 0.01
wp = torch.empty(9 * 64 * 64, device=dev)
_lib.check(L.pnp_conv3x3_pack_f16x3(s, p(w), p(wp), 64))
flop = 2.0 * n * H * W * 64 * 64 * 9
for data in ('random', 'zeros'):
    x = torch.relu(torch.randn(n, H, W, 64, device=dev)) if data == 'random' else torch.zeros(n, H, W, 64, device=dev)
    xs = D.split_activations(x)
    y = torch.empty_like(x)
    for fmt, tag in ((0, 'float32 -> float32'), (4, 'float32 -> split  '), (5, 'split   -> split  '), (1, 'split   -> float32')):
        if only is not None and fmt != only:
            continue
        xin = xs if fmt & 1 else x
        run = lambda: _lib.check(L.pnp_conv3x3_nhwc_f16x3_fmt(s, p(xin), p(wp), p(b), None, p(y), n, 64, H, W, 1, 1, fmt))
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 40
        print('%-7s %s  %.4f ms  %.1f TFLOP/s float32-equivalent  %.3f of the 2.5 PFLOP/s f16 peak (3 products issued)' % (data, tag, ms, flop / ms / 1e9, 3 * flop / ms / 1e9 / 2.5e6), flush=True)
