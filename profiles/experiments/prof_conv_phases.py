#!/usr/bin/env python3
"""Phase clocks of the conv kernel (-DCV_PROF build: bash profiles/variants.sh build kernels_conv.hip prof "-DCV_PROF"):
shader-clock sums of wave 0 of every workgroup over ONE launch at [64, 64, 128, 128].
usage (GPU box): PNP_MRI_LIB=build/variants/lib_prof.so python3 profiles/experiments/prof_conv_phases.py"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pnp_admm_cnc_mri_amd import _lib
L = _lib.lib()
n, H, W = 64, 128, 128
x = torch.randn(n, H, W, 64, device='cuda'); y = torch.empty_like(x)
w = torch.randn(64, 64, 3, 3, device='cuda') * 0.06; b = torch.zeros(64, device='cuda'); pk = torch.empty(9 * 64 * 64, device='cuda')
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(L.pnp_conv3x3_c64_pack(s, C.c_void_p(w.data_ptr()), C.c_void_p(pk.data_ptr())))
for _ in range(5):
    _lib.check(L.pnp_conv3x3_c64_nhwc(s, C.c_void_p(x.data_ptr()), C.c_void_p(pk.data_ptr()), C.c_void_p(b.data_ptr()), None, C.c_void_p(y.data_ptr()), n, H, W, 1, 1))
torch.cuda.synchronize()
out = np.zeros((1024, 8), np.uint64)
raw = C.CDLL(_lib.LIB_PATH)
assert raw.pnp_conv_prof_read(out.ctypes.data_as(C.c_void_p)) == 0
p = out[:512].astype(np.float64)
names = ['loop top', 'issue next-tile loads', 'nine taps (MFMA)', 'barrier after taps', 'epilogue', 'barrier before hand-over', 'hand-over + barrier', '-']
tot = p.sum(axis=1)
print('tiles per workgroup: %d; cycles per workgroup (median) %.0f' % (n * (H // 8) * (W // 16) // 512, np.median(tot)))
for k in range(7):
    print('%-28s median %9.0f cycles per launch = %5.1f %%   (per tile %7.0f)' % (names[k], np.median(p[:, k]), 100 * np.median(p[:, k]) / np.median(tot), np.median(p[:, k]) / 16))
