#!/usr/bin/env python3
"""MIOpen's fp32 conv3x3 (torch, find mode on, channels_last as the Denoiser runs it) on the layer shapes of DRUNet at 64 slices of
256 x 256 per call: where would a kernel of our own have something to gain?   usage (GPU box): python3 profiles/experiments/probe_miopen_shapes.py"""
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = 'cuda'


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


for n, c, hw in ((64, 64, 256), (64, 128, 128), (64, 256, 64), (64, 512, 32), (16, 64, 256), (16, 128, 128), (16, 256, 64), (16, 512, 32)):
    x = torch.randn(n, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(c, c, 3, 3, device=dev) * (2.0 / (9 * c)) ** 0.5).contiguous(memory_format=torch.channels_last)
    t = timeit(lambda: F.conv2d(x, w, None, padding=1))
    flop = 2.0 * n * hw * hw * c * c * 9
    print('conv3x3 %3d -> %3d at [%d, %d, %d]: %.3f ms  %.1f TFLOP/s  %.3f of 157.3' % (c, c, n, hw, hw, t, flop / t / 1e9, flop / t / 1e9 / 157.3))
for n, ci, co, hw in ((64, 64, 128, 256), (64, 128, 256, 128), (64, 256, 512, 64)):
    x = torch.randn(n, ci, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 2, 2, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    t = timeit(lambda: F.conv2d(x, w, None, stride=2))
    flop = 2.0 * n * (hw // 2) ** 2 * ci * co * 4
    print('strideconv 2x2 %3d -> %3d at [%d, %d, %d]: %.3f ms  %.1f TFLOP/s' % (ci, co, n, hw, hw, t, flop / t / 1e9))
    x2 = torch.randn(n, co, hw // 2, hw // 2, device=dev).contiguous(memory_format=torch.channels_last)
    w2 = (torch.randn(co, ci, 2, 2, device=dev) * 0.05)
    t = timeit(lambda: F.conv_transpose2d(x2, w2, None, stride=2))
    print('convtranspose 2x2 %3d -> %3d at [%d, %d, %d]: %.3f ms  %.1f TFLOP/s' % (co, ci, n, hw // 2, hw // 2, t, flop / t / 1e9))
