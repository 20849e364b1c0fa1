#!/usr/bin/env python3
"""Phase clocks of the f16x3 conv kernel (-DH3_PROF build: bash profiles/variants.sh build kernels_conv_f16x3.hip h3prof "-DH3_PROF"):
shader-clock sums of wave 0 of every workgroup over ONE launch.
usage (GPU box): PNP_MRI_LIB=build/variants/lib_h3prof.so python3 profiles/experiments/prof_conv_f16x3_phases.py [n C H W]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pnp_admm_cnc_mri_amd import _lib
L = _lib.lib()
n, ch, H, W = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (64, 64, 128, 128)
x = torch.randn(n, H, W, ch, device='cuda'); y = torch.empty_like(x)
w = torch.randn(ch, ch, 3, 3, device='cuda') * (2.0 / (9 * ch)) ** 0.5; pk = torch.empty(9 * ch * ch, device='cuda')
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
_lib.check(L.pnp_conv3x3_pack_f16x3(s, p(w), p(pk), ch))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for k in range(5):
    if k == 4:
        e0.record()
    _lib.check(L.pnp_conv3x3_nhwc_f16x3(s, p(x), p(pk), None, None, p(y), n, ch, H, W, 1))
e1.record()
torch.cuda.synchronize()
out = np.zeros((1024, 8), np.uint64)
raw = C.CDLL(_lib.LIB_PATH)
assert raw.pnp_conv_h3_prof_read(out.ctypes.data_as(C.c_void_p)) == 0
items = n * ((H + 7) // 8) * ((W + 15) // 16) * (ch // 64)
wgs = min(512, items)
q = out[:wgs].astype(np.float64)
names = ['loop top', 'issue next-input loads', 'tap barriers', 'weights: LDS write + next request', 'taps: LDS reads + MFMA', 'barrier after taps',
         'epilogue', 'barrier + hand-over']
tot = q.sum(axis=1)
per = items / wgs * (ch // 64)
print('[%d, %d, %d, %d]: %.3f ms (instrumented); chunks (9 taps each) per workgroup %.1f; cycles per workgroup (median) %.0f = %.2f GHz'
      % (n, ch, H, W, e0.elapsed_time(e1), per, np.median(tot), np.median(tot) / e0.elapsed_time(e1) / 1e6))
for k in range(8):
    print('%-36s median %9.0f cycles per launch = %5.1f %%   (per chunk %7.0f)' % (names[k], np.median(q[:, k]), 100 * np.median(q[:, k]) / np.median(tot), np.median(q[:, k]) / per))
print('MFMA work of one wave per chunk: 216 x 32 = 6912 cycles')
