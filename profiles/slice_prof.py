#!/usr/bin/env python3
"""Reads the phase clock dump of the slice-resident kernel (PNP_SLICE_PROF=<file>) and prints the
median duration of each phase in microseconds (wall_clock64 ticks at 100 MHz)."""
import sys
import numpy as np
raw = open(sys.argv[1], 'rb').read()
B, iters = np.frombuffer(raw[:8], np.int32)
t = np.frombuffer(raw[8:], np.int64).reshape(B, 2 + 6 * iters).astype(np.float64) / 100.0   # us
t = t[t[:, 0] > 0]
print('workgroups with a record:', len(t), 'iterations', iters)
first = t[:, 1] - t[:, 0]
ph = t[:, 2:].reshape(len(t), iters, 6)
prev = np.concatenate([t[:, 1:2], ph[:, :-1, 5]], axis=1)
names = ['wait(rows)', 'T1', 'columns', 'wait(cols)', 'T2', 'rows']
print('rows(first)  median %.2f us' % np.median(first))
for k, n in enumerate(names):
    d = ph[:, :, k] - (prev if k == 0 else ph[:, :, k - 1])
    print('%-10s   median %.2f us   p10 %.2f  p90 %.2f' % (n, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
print('iteration    median %.2f us' % np.median(ph[:, :, 5] - prev))
t0 = t[:, 0].min()
start, end = t[:, 0] - t0, t[:, -1] - t0
dur = end - start
print('workgroup duration: median %.1f us  p10 %.1f  p90 %.1f  max %.1f;  makespan %.1f us' % (
    np.median(dur), np.percentile(dur, 10), np.percentile(dur, 90), dur.max(), end.max()))
order = np.argsort(start)
first_round = start < np.median(dur) * 0.5
print('workgroups starting in the first half-duration: %d; their end: median %.1f  max %.1f;  the others start: min %.1f median %.1f max %.1f' % (
    first_round.sum(), np.median(end[first_round]), end[first_round].max(),
    start[~first_round].min() if (~first_round).any() else -1, np.median(start[~first_round]) if (~first_round).any() else -1,
    start[~first_round].max() if (~first_round).any() else -1))
per_it = (ph[:, :, 5] - prev)
print('per-iteration time by iteration index (median over workgroups):', np.round(np.median(per_it, axis=0)[:12], 1))
print('  first-round workgroups %.1f us/iteration, second-round %.1f' % (np.median(per_it[first_round]), np.median(per_it[~first_round]) if (~first_round).any() else -1))

# which workgroups are slow?  blocks b and b + 8 share an XCD (round-robin dispatch): per-residue medians
raw_all = np.frombuffer(raw[8:], np.int64).reshape(B, 2 + 6 * iters).astype(np.float64) / 100.0
d_all = raw_all[:, -1] - raw_all[:, 0]
for rnd, sel in (('first 256 blocks', slice(0, 256)), ('blocks 256..', slice(256, None))):
    dd = d_all[sel]
    if len(dd):
        print(rnd, 'duration by blockIdx %% 8:', np.round([np.median(dd[r::8]) for r in range(8)], 0))

# per-phase medians by blockIdx % 8 for the first round (one workgroup per compute unit, all started together):
# a phase that touches no memory (T1, T2) separates clock differences between XCDs from memory-path differences
ph_all = raw_all[:, 2:].reshape(B, iters, 6)
prev_all = np.concatenate([raw_all[:, 1:2], ph_all[:, :-1, 5]], axis=1)
nfirst = min(256, B)
print('first-round medians by blockIdx %% 8 (us): wait(rows) T1 columns wait(cols) T2 rows | iteration')
for r in range(8):
    sel = np.arange(r, nfirst, 8)
    row = []
    for k in range(6):
        d = ph_all[sel, :, k] - (prev_all[sel] if k == 0 else ph_all[sel, :, k - 1])
        row.append(np.median(d))
    it = np.median(ph_all[sel, :, 5] - prev_all[sel])
    print('  %d: ' % r + ' '.join('%6.2f' % v for v in row) + ' | %6.2f' % it)

# second round (blocks 256..): does its pace converge to the first round's?  per-iteration time by iteration index
if B > 256:
    d2 = (ph_all[256:, :, 5] - prev_all[256:])
    d1 = (ph_all[:256, :, 5] - prev_all[:256])
    idx = list(range(0, min(iters, 12))) + list(range(12, max(12, iters - 4), max(1, (iters - 16) // 16))) + list(range(max(12, iters - 4), iters))
    print('iteration index            :', idx)
    print('first round  (median, us)  :', np.round(np.median(d1, axis=0)[idx], 1))
    print('second round (median, us)  :', np.round(np.median(d2, axis=0)[idx], 1))
    print('first round, even / odd blocks:', np.round(np.median(d1[0::2], axis=0)[idx], 1), '/', np.round(np.median(d1[1::2], axis=0)[idx], 1))
    for k, nm in ((1, 'T1'), (2, 'columns'), (4, 'T2'), (5, 'rows')):
        dk = ph_all[:256, :, k] - ph_all[:256, :, k - 1]
        print('first round %-8s by index :' % nm, np.round(np.median(dk, axis=0)[idx], 2))
        dk2 = ph_all[256:, :, k] - ph_all[256:, :, k - 1]
        print('second round %-7s by index :' % nm, np.round(np.median(dk2, axis=0)[idx], 2))
    rows2 = ph_all[256:, :, 5] - ph_all[256:, :, 4]
    rows1 = ph_all[:256, :, 5] - ph_all[:256, :, 4]
    print('row phase first / second round (median over all iterations): %.2f / %.2f us' % (np.median(rows1), np.median(rows2)))
    st = raw_all[256:, 0] - raw_all[:, 0].min()
    print('second-round start times: min %.0f median %.0f max %.0f us; first-round end: min %.0f median %.0f max %.0f' % (
        st.min(), np.median(st), st.max(), (raw_all[:256, -1] - raw_all[:, 0].min()).min(), np.median(raw_all[:256, -1] - raw_all[:, 0].min()), (raw_all[:256, -1] - raw_all[:, 0].min()).max()))

# -DSLICE_PROF_CLOCK build: slot k = 1 of every iteration holds the shader clock (s_memtime): MHz the compute unit ran at, by iteration
if len(sys.argv) > 2 and sys.argv[2] == 'clock':
    cyc = ph_all[:, 1:, 1] - ph_all[:, :-1, 1]                 # shader cycles between the T1 stamps of consecutive iterations (x 100: undo the /100)
    wall = ph_all[:, 1:, 5] - ph_all[:, :-1, 5]                # us between the ends of consecutive iterations
    mhz = cyc * 100.0 / wall
    step = max(1, (iters - 1) // 24)
    ii = list(range(0, iters - 1, step))
    print('iteration index               :', ii)
    print('first round  clock (MHz)      :', np.round(np.median(mhz[:256], axis=0)[ii]))
    if B > 256:
        print('second round clock (MHz)      :', np.round(np.median(mhz[256:], axis=0)[ii]))
    print('first round  iteration (us)   :', np.round(np.median(wall[:256], axis=0)[ii], 1))
    print('first round  cycles/iteration :', np.round(np.median(cyc[:256] * 100.0, axis=0)[ii]))
