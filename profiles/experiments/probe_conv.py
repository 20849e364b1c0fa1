#!/usr/bin/env python3
"""Probe of the HIP conv3x3 (64 -> 64, fp32 MFMA implicit GEMM, kernels_conv.hip) against MIOpen (torch.nn.functional.conv2d)
on the shape of FFDNet's body layers at 64 images per call: [64, 64, 128, 128].  Correctness (rel-L2 of one layer and of a
13-layer conv + ReLU chain) and time per layer.   usage (GPU box): python3 profiles/experiments/probe_conv.py [n H W [dilation [f32|f16x3 [input scale]]]]"""
import ctypes as C
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pnp_admm_cnc_mri_amd import _lib  # noqa: E402

n, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 128, 128)
DIL = int(sys.argv[4]) if len(sys.argv) >= 5 else 1                # dilation (= zero padding) of the layer
MATH = sys.argv[5] if len(sys.argv) >= 6 else 'f32'                # 'f32' (kernels_conv.hip) or 'f16x3' (kernels_conv_f16x3.hip)
SCALE = float(sys.argv[6]) if len(sys.argv) >= 7 else 1.0          # input magnitude (f16x3: small values exercise the lo halves' scaling)
L = _lib.lib()
dev = torch.device('cuda', 0)
torch.manual_seed(0)
x = torch.randn(n, 64, H, W, device=dev) * SCALE
ws = [torch.randn(64, 64, 3, 3, device=dev) * (2.0 / 576) ** 0.5 for _ in range(13)]
bs = [torch.randn(64, device=dev) * 0.01 for _ in range(13)]
stream = torch.cuda.current_stream().cuda_stream
wt = []
for w in ws:                                                       # packed once per layer into the kernel's fragment order
    pk = torch.empty(9 * 64 * 64, device=dev)
    _lib.check((L.pnp_conv3x3_c64_pack if MATH == 'f32' else L.pnp_conv3x3_c64_pack_f16x3)(C.c_void_p(stream), C.c_void_p(w.data_ptr()), C.c_void_p(pk.data_ptr())))
    wt.append(pk)


def hip_conv(xn, w, b, relu, out):
    _lib.check((L.pnp_conv3x3_c64_nhwc if MATH == 'f32' else L.pnp_conv3x3_c64_nhwc_f16x3)(C.c_void_p(stream), C.c_void_p(xn.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()),
                                      None, C.c_void_p(out.data_ptr()), n, H, W, 1 if relu else 0, DIL))
    return out


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


xn = torch.empty(n, H, W, 64, device=dev)
_lib.check(L.pnp_relayout_c64(C.c_void_p(stream), C.c_void_p(x.data_ptr()), C.c_void_p(xn.data_ptr()), n, H, W, 1))
assert torch.equal(xn, x.permute(0, 2, 3, 1).contiguous())
back = torch.empty_like(x)
_lib.check(L.pnp_relayout_c64(C.c_void_p(stream), C.c_void_p(xn.data_ptr()), C.c_void_p(back.data_ptr()), n, H, W, 0))
assert torch.equal(back, x)

torch.backends.cudnn.benchmark = True
ref1 = F.relu(F.conv2d(x, ws[0], bs[0], padding=DIL, dilation=DIL))
y1 = hip_conv(xn, wt[0], bs[0], True, torch.empty_like(xn))
ref64 = F.relu(F.conv2d(x.double(), ws[0].double(), bs[0].double(), padding=DIL, dilation=DIL))
print('one layer: hip vs torch fp32 %.3e | hip vs fp64 %.3e | torch fp32 vs fp64 %.3e' %
      (rel(y1.permute(0, 3, 1, 2), ref1), rel(y1.permute(0, 3, 1, 2), ref64), rel(ref1, ref64)))
# chain of 13 conv + ReLU
a, b_ = xn, torch.empty_like(xn)
r = x
for k in range(13):
    hip_conv(a, wt[k], bs[k], True, b_)
    a, b_ = b_, (a if a is not xn else torch.empty_like(xn))
    r = F.relu(F.conv2d(r, ws[k], bs[k], padding=DIL, dilation=DIL))
print('13 layers: hip vs torch fp32 %.3e' % rel(a.permute(0, 3, 1, 2), r))


def timeit(fn, reps=20):
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


flop = 2.0 * n * H * W * 64 * 64 * 9
o = torch.empty_like(xn)
t_hip = timeit(lambda: hip_conv(xn, wt[0], bs[0], True, o))
t_mi = timeit(lambda: F.relu_(F.conv2d(x, ws[0], bs[0], padding=DIL, dilation=DIL)))
xcl = x.contiguous(memory_format=torch.channels_last)
wcl = ws[0].contiguous(memory_format=torch.channels_last)
t_cl = timeit(lambda: F.relu_(F.conv2d(xcl, wcl, bs[0], padding=DIL, dilation=DIL)))
for name, t in (('hip mfma ' + MATH, t_hip), ('miopen nchw', t_mi), ('miopen channels_last', t_cl)):
    print('%-22s %.3f ms  %.1f TFLOP/s  %.3f of 157.3' % (name, t, flop / t / 1e9, flop / t / 1e9 / 157.3))
