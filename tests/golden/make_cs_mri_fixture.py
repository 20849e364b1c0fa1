#!/usr/bin/env python3
"""Writes tests/golden/cs_mri_fixture/*.mat -- the reference's CS_MRI directory layout (Q1 / Q11 /
OMEGA masks, noises) rebuilt with scipy.io.savemat from the arrays already committed in
tests/golden/inputs_set1_05.npz (data only; oracle/make_golden.py checks those arrays against the
reference's own files where they are readable)."""
import os
import sys

import numpy as np
import scipy.io as sio

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from pnp_admm_cnc_mri_amd import imageio as IO   # noqa: E402

d = np.load(os.path.join(HERE, 'inputs_set1_05.npz'))
out = os.path.join(HERE, 'cs_mri_fixture')
os.makedirs(out, exist_ok=True)
for name in IO.MASK_NAMES:
    q1 = np.unpackbits(d[name + '_packbits'])[:65536].reshape(256, 256)
    IO.save_mask_mat(os.path.join(out, name + '.mat'), q1, with_omega=(name != 'Q_Radial30'))   # as the reference's files
sio.savemat(os.path.join(out, 'noises.mat'), {'noises': d['noises_c128']}, do_compression=True)
print(sorted(os.listdir(out)), sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out)))
