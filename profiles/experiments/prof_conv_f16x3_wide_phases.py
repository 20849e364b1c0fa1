#!/usr/bin/env python3
"""Phase clocks of the WIDE f16x3 conv kernel (-DH3W_PROF build: bash profiles/variants.sh build kernels_conv_f16x3_wide.hip h3wprof "-DH3W_PROF"):
shader-clock sums of compute wave 0 of every workgroup over ONE launch.
usage (GPU box): PNP_CONV_WIDE=1 PNP_MRI_LIB=build/variants/lib_h3wprof.so python3 profiles/experiments/prof_conv_f16x3_wide_phases.py [n C H W fmt]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pnp_admm_cnc_mri_amd import _lib, denoisers as D
L = _lib.lib()
n, ch, H, W, fmt = (int(v) for v in sys.argv[1:6]) if len(sys.argv) >= 6 else (64, 64, 128, 128, 5)
x = torch.relu(torch.randn(n, H, W, ch, device='cuda')); y = torch.empty_like(x)
xin = D.split_activations(x) if fmt & 1 else x
w = torch.randn(ch, ch, 3, 3, device='cuda') * (2.0 / (9 * ch)) ** 0.5; pk = torch.empty(9 * ch * ch, device='cuda')
b = torch.randn(ch, device='cuda') * 0.1
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
_lib.check(L.pnp_conv3x3_pack_f16x3(s, p(w), p(pk), ch))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for k in range(5):
    if k == 4:
        e0.record()
    _lib.check(L.pnp_conv3x3_nhwc_f16x3_fmt(s, p(xin), p(pk), p(b), None, p(y), n, ch, H, W, 1, 1, fmt))
e1.record()
torch.cuda.synchronize()
out = np.zeros((1024, 8), np.uint32)
raw = C.CDLL(_lib.LIB_PATH)
assert raw.pnp_conv_h3w_prof_read(out.ctypes.data_as(C.c_void_p)) == 0
NC = ch // 64
items = n * ((H + 15) // 16) * ((W + 15) // 16) * NC
wgs = min(256 - 256 % NC, items)
q = out[:wgs].astype(np.float64)
names = ['between chunks (besides barriers, epilogue)', 'waiting at B_0 (input hand-over)', 'taps: LDS reads + MFMA', 'waiting at B_1..8', 'waiting at E', 'epilogue', '-', '-']
tot = q.sum(axis=1)
per = items / wgs * 1.0                                 # items per workgroup
print('[%d, %d, %d, %d] fmt %d: %.3f ms (instrumented); items per workgroup %.1f (x %d chunks of 9 taps); cycles per workgroup (median) %.0f = %.2f GHz'
      % (n, ch, H, W, fmt, e0.elapsed_time(e1), per, NC, np.median(tot), np.median(tot) / e0.elapsed_time(e1) / 1e6))
for k in range(6):
    print('%-46s median %9.0f cycles per launch = %5.1f %%   (per chunk %7.0f)' % (names[k], np.median(q[:, k]), 100 * np.median(q[:, k]) / np.median(tot), np.median(q[:, k]) / per / NC))
print('MFMA work of one compute wave per chunk: 864 x 16 = 13824 cycles')
