#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_kt, prof_fetch, prof_write) into the small tracked
files under profiles/: kernel stats of the pnp:: kernels and per-launch HBM traffic.

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies 128-B requests at 64 B
(MI355X_MICROARCH.md, HBM section), so read bytes = 2 * FETCH_SIZE * 1024; the correction is
confirmed here on kernels whose byte counts are known exactly (k_frows<first>: reads z and w only,
2 * 131,127 KiB = 268.5 MB = 2 arrays x 512 slices x 256 KiB).
usage: python profiles/summarize.py <round-tag>
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'gpurun_out')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
update_traffic = '--no-traffic' not in sys.argv


def short(name):
    name = name.replace('pnp::', '').replace('void ', '')
    return name.split('(')[0]


stats = []
f = max(glob.glob(os.path.join(G, 'prof_kt', '*', '*_kernel_stats.csv')), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    if 'pnp::' in r['Name']:
        stats.append({'kernel': short(r['Name']), 'calls': int(r['Calls']), 'avg_us': float(r['AverageNs']) / 1e3,
                      'min_us': float(r['MinNs']) / 1e3, 'max_us': float(r['MaxNs']) / 1e3,
                      'total_ms': float(r['TotalDurationNs']) / 1e6, 'pct': float(r['Percentage'])})
pmc = collections.defaultdict(dict)
for name, ctr in (('prof_fetch', 'FETCH_SIZE'), ('prof_write', 'WRITE_SIZE')):
    fs = glob.glob(os.path.join(G, name, '*', '*_counter_collection.csv'))
    if not fs:
        continue
    acc = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        if 'pnp::' in r['Kernel_Name'] and r['Counter_Name'] == ctr:
            acc[short(r['Kernel_Name'])].append(float(r['Counter_Value']))
            meta[short(r['Kernel_Name'])] = (int(r['VGPR_Count']), int(r['LDS_Block_Size']), int(r['Grid_Size']))
    for k, v in acc.items():
        pmc[k][ctr + '_KiB_avg'] = sum(v) / len(v)
        pmc[k]['launches_' + ctr] = len(v)
        pmc[k]['vgpr'], pmc[k]['lds_bytes'], pmc[k]['grid_threads'] = meta[k]
for k, d in pmc.items():
    rd = 2.0 * d.get('FETCH_SIZE_KiB_avg', 0) * 1024
    wr = d.get('WRITE_SIZE_KiB_avg', 0) * 1024
    d['read_bytes_corrected'] = rd
    d['write_bytes'] = wr
    d['hbm_bytes_per_launch'] = rd + wr
bench = None
for line in open(os.path.join(G, 'prof_kt.log')):
    if line.startswith('{"metric"'):
        bench = json.loads(line)
out = {'tag': tag, 'command': 'rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline '
                              '(+ separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes with --steps 20 --warmup 2)',
       'kernel_stats': stats, 'pmc': pmc, 'bench_line_under_profiler': bench}
iter_kernels = [s for s in stats if s['calls'] >= 50]
steps = (bench or {}).get('steps', 100) + (bench or {}).get('warmup', 10)
out['per_iteration'] = {
    'kernels': [s['kernel'] for s in iter_kernels],
    # launches of each loop kernel per batched iteration (2 for the sequential schedule; 4 mixed
    # launches with two queues, which overlap pairwise)
    'launches_per_iteration': {s['kernel']: round(s['calls'] / steps, 2) for s in iter_kernels},
    'sum_kernel_time_us': sum(s['avg_us'] * s['calls'] / steps for s in iter_kernels),
    'wall_us_per_iteration_hip_events': (bench or {}).get('hip_event_ms_per_step', 0) * 1e3,
    'hbm_bytes': sum(pmc.get(s['kernel'], {}).get('hbm_bytes_per_launch', 0) * s['calls'] / steps for s in iter_kernels),
    'algorithmic_bytes_57N': 57 * 65536 * 512,
}
json.dump(out, open(os.path.join(ROOT, 'profiles', 'rocprof_%s.json' % tag), 'w'), indent=1)
with open(os.path.join(ROOT, 'profiles', 'rocprof_%s_kernel_stats.csv' % tag), 'w') as fo:
    fo.write(open(f).read())
print(json.dumps(out['per_iteration'], indent=1))
for s in stats:
    d = pmc.get(s['kernel'], {})
    print('%-40s calls %4d avg %8.1f us  hbm %7.1f MB  -> %5.2f TB/s' % (
        s['kernel'], s['calls'], s['avg_us'], d.get('hbm_bytes_per_launch', 0) / 1e6,
        d.get('hbm_bytes_per_launch', 0) / (s['avg_us'] * 1e-6) / 1e12 if s['avg_us'] else 0))
# bench.py reads this for roofline.traffic
if not update_traffic:
    sys.exit(0)
tj = os.path.join(ROOT, 'profiles', 'traffic.json')
cur = json.load(open(tj)) if os.path.exists(tj) else {}
path = bench['config']['path'] if bench else 'fused'
cur[path] = {'hbm_bytes_per_iteration_b512': out['per_iteration']['hbm_bytes'], 'from': 'profiles/rocprof_%s.json' % tag}
json.dump(cur, open(tj, 'w'), indent=1)
