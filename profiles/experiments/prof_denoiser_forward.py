#!/usr/bin/env python3
"""One denoiser forward at 64 slices, kernel by kernel, WITHOUT the MIOpen find trials of the warm-up: run under
rocprofv3 --kernel-trace; a marker launch (k_relayout64 on a 1 x 3 image) separates warm-up from the measured forwards.
usage (GPU box): cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 $R/profiles/experiments/prof_denoiser_forward.py MODEL BACKEND [size]
then: python3 $R/profiles/experiments/prof_denoiser_forward.py --summarize OUT"""
import csv, ctypes as C, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REPS = 4
if sys.argv[1] == '--summarize':
    rows = []
    for f in glob.glob(sys.argv[2] + '/**/*kernel_trace.csv', recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'k_relayout64' in r['Kernel_Name'] and int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0) <= 256]
    rows = rows[marks[-1] + 1:]
    tot, acc = 0.0, {}
    for r in rows:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        a = acc.setdefault(r['Kernel_Name'][:100], [0.0, 0]); a[0] += d; a[1] += 1; tot += d
    span = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3
    print('%d forwards: %.2f ms of kernels per forward, %.2f ms wall per forward' % (REPS, tot / REPS / 1e3, span / REPS / 1e3))
    for k, (d, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:16]:
        print('%-100s %5.1f calls/fwd  %8.1f us each  %7.3f ms/fwd %5.1f%%' % (k, n / REPS, d / n, d / REPS / 1e3, 100 * d / tot))
    sys.exit(0)
import torch
from pnp_admm_cnc_mri_amd import _lib, denoisers as D
name, backend = sys.argv[1], sys.argv[2]
size = int(sys.argv[3]) if len(sys.argv) > 3 else 256
net, nlm, _ = D.build(name)
net.load_state_dict(D.seeded_state_dict(net, 1))
sig = torch.tensor([30.0 / 255]) if name.startswith(('drunet', 'ircnn')) else None
den = D.Denoiser(name, net.eval(), nlm, sigmas=sig, backend=backend).to('cuda')
x = torch.rand(64 if size == 256 else 16, 1, size, size, device='cuda')
for _ in range(2):
    den(x, 0)
torch.cuda.synchronize()
L = _lib.lib()
a, b = torch.zeros(1, 64, 1, 3, device='cuda'), torch.zeros(1, 1, 3, 64, device='cuda')
_lib.check(L.pnp_relayout_c64(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), 1, 1, 3, 1))
for _ in range(REPS):
    den(x, 0)
torch.cuda.synchronize()
