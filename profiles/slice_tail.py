#!/usr/bin/env python3
"""Tail of a multi-round launch of the slice-resident kernel, from a -DSLICE_PROF dump (PNP_SLICE_PROF=<file>):
how much of the launch is compute units standing idle at its end, and who the late ones are."""
import sys
import numpy as np
raw = open(sys.argv[1], 'rb').read()
B, iters = np.frombuffer(raw[:8], np.int32)
t = np.frombuffer(raw[8:], np.int64).reshape(B, 2 + 6 * iters).astype(np.float64) / 100.0   # us
t0 = t[:, 0].min()
start, end = t[:, 0] - t0, t[:, -1] - t0
dur = end - start
cus = min(256, B)
print('slices %d, iterations %d: makespan %.1f us, sum of workgroup durations / %d units = %.1f us -> idle tail %.1f %%' % (
    B, iters, end.max(), cus, dur.sum() / cus, 100 * (1 - dur.sum() / cus / end.max())))
idx = np.arange(B)
for name, sel in (('first %d blocks' % cus, idx < cus), ('later blocks', idx >= cus)):
    if sel.any():
        for par in (0, 1):
            s = sel & ((idx & 1) == par)
            print('  %-16s %s slices: duration median %.1f (p10 %.1f p90 %.1f)  start median %.1f  end median %.1f max %.1f' % (
                name, 'odd ' if par else 'even', np.median(dur[s]), np.percentile(dur[s], 10), np.percentile(dur[s], 90), np.median(start[s]), np.median(end[s]), end[s].max()))
if B > cus:
    # which first-round workgroup did each later workgroup follow?  (a compute unit runs one at a time: match start to the nearest earlier end)
    e1 = np.sort(end[:cus]); s2 = np.sort(start[cus:])
    print('  first-round ends  : min %.1f p25 %.1f median %.1f p75 %.1f max %.1f' % (e1.min(), np.percentile(e1, 25), np.median(e1), np.percentile(e1, 75), e1.max()))
    print('  later-round starts: min %.1f p25 %.1f median %.1f p75 %.1f max %.1f' % (s2.min(), np.percentile(s2, 25), np.median(s2), np.percentile(s2, 75), s2.max()))
    order = np.argsort(start[cus:]) + cus
    print('  later blocks in start order (blockIdx): first 16', order[:16], ' last 16', order[-16:])
    late = np.argsort(end)[-16:]
    print('  the 16 workgroups that end last: blockIdx', late, 'durations', np.round(dur[late], 0), 'starts', np.round(start[late], 0))
    print('  correlation(start, duration) of later blocks: %.2f' % np.corrcoef(start[cus:], dur[cus:])[0, 1])
