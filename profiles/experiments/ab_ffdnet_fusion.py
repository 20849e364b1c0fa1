#!/usr/bin/env python3
"""A/B on one box: bench_pnp.py's FFDNet line with FFDNet's input / output stages folded into its first / last layer (the product) and
with them as PyTorch launches (pad, pixel-unshuffle, concatenation, pixel-shuffle, crop, copy into the result: rounds 2-4).
usage (GPU box): python3 profiles/experiments/ab_ffdnet_fusion.py fused|unfused [bench_pnp args]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mode = sys.argv[1]
sys.argv = [os.path.join(ROOT, 'bench_pnp.py')] + sys.argv[2:]
from pnp_admm_cnc_mri_amd import denoisers as D
if mode == 'unfused':
    D.FFDNet._fused_ok = lambda self, x, sigma: False
import bench_pnp
bench_pnp.main()
