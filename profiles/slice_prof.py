#!/usr/bin/env python3
"""Reads the phase clock dump of the slice-resident kernel (PNP_SLICE_PROF=<file>) and prints the
median duration of each phase in microseconds (wall_clock64 ticks at 100 MHz)."""
import sys
import numpy as np
raw = open(sys.argv[1], 'rb').read()
B, iters = np.frombuffer(raw[:8], np.int32)
t = np.frombuffer(raw[8:], np.int64).reshape(B, 2 + 4 * iters).astype(np.float64) / 100.0   # us
t = t[t[:, 0] > 0]
print('workgroups with a record:', len(t), 'iterations', iters)
first = t[:, 1] - t[:, 0]
ph = t[:, 2:].reshape(len(t), iters, 4)
prev = np.concatenate([t[:, 1:2], ph[:, :-1, 3]], axis=1)
names = ['T1', 'columns', 'T2', 'rows']
print('rows(first)  median %.2f us' % np.median(first))
for k, n in enumerate(names):
    d = ph[:, :, k] - (prev if k == 0 else ph[:, :, k - 1])
    print('%-10s   median %.2f us   p10 %.2f  p90 %.2f' % (n, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
print('iteration    median %.2f us' % np.median(ph[:, :, 3] - prev))
print('start spread of workgroups %.1f us; total span %.1f us' % (t[:, 0].max() - t[:, 0].min(), t[:, -1].max() - t[:, 0].min()))
