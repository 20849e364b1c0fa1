#!/usr/bin/env python3
"""Probe: does a captured HIP graph (torch.cuda.CUDAGraph) speed up the single-image denoiser forward?  (launch-bound at batch 1)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pnp_admm_cnc_mri_amd import denoisers as D

dev = torch.device('cuda', 0)
for name in sys.argv[1:] or ['ffdnet_gray', 'drunet_gray', 'dncnn_25']:
    for B in (1, 4):
        try:
            net, nlm, sched = D.build(name)
        except Exception as e:
            print(name, 'build failed', e); break
        net.load_state_dict(D.seeded_state_dict(net, 1))
        sig = torch.full((64,), 0.05) if sched else None
        den = D.Denoiser(name, net.eval(), nlm, sigmas=sig, noises=torch.zeros(256, 256).numpy() if D.family(name) == 'fdncnn' else None,
                         cnn_batch=B, miopen_find=True).to(dev)
        x = torch.rand((B, 1, 256, 256), device=dev)
        out = torch.empty_like(x)
        with torch.no_grad():
            for _ in range(3): den(x, 0, out=out)          # MIOpen find + warm-up outside any capture
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(20): den(x, 0, out=out)
            torch.cuda.synchronize()
            eager = (time.perf_counter() - t0) / 20 * 1e3
            ref = out.clone()
            den.miopen_find = False
            torch.backends.cudnn.benchmark = True          # keep the algorithms found above
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            try:
                with torch.cuda.stream(s):
                    den(x, 0, out=out)
                torch.cuda.current_stream().wait_stream(s)
                with torch.cuda.graph(g):
                    den(x, 0, out=out)
                torch.cuda.synchronize()
                out.zero_()
                t0 = time.perf_counter()
                for i in range(20): g.replay()
                torch.cuda.synchronize()
                graphed = (time.perf_counter() - t0) / 20 * 1e3
                print('%s batch %d: eager %.3f ms  graph %.3f ms  (x%.2f)  max|diff| %.2e' % (name, B, eager, graphed, eager / graphed, float((out - ref).abs().max())))
            except Exception as e:
                print('%s batch %d: eager %.3f ms  capture failed: %s' % (name, B, eager, str(e)[:200]))
