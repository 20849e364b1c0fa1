#!/usr/bin/env python3
"""Condense rocprofv3 output of profiles/collect.sh (gpurun_out/prof_<name>/{kt,fetch,write}) into the small
tracked files under profiles/: per-kernel time and HBM traffic of the pnp:: kernels, per-iteration totals,
and the entry of profiles/traffic.json that bench.py reports as roofline.traffic.

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies 128-B requests at 64 B
(MI355X_MICROARCH.md, HBM section), so read bytes = 2 * FETCH_SIZE * 1024; the correction is confirmed on
kernels whose byte counts are known exactly (k_frows<first>: reads z and w only, 2 * 131,127 KiB = 268.5 MB
= 2 arrays x 512 slices x 256 KiB).
usage: python profiles/summarize.py <round-tag> <name> [<name> ...]
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'gpurun_out')
LOOP = ('k_frows', 'k_fcols', 'k_fmixed', 'k5_rows', 'k5_cols', 'k_slice')     # the generic path's loop kernels: by call count


def short(name):
    return name.replace('pnp::', '').replace('void ', '').split('(')[0]


def newest(pattern):
    fs = glob.glob(pattern, recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None


def one(tag, name):
    d = os.path.join(G, 'prof_' + name)
    stats = []
    f = newest(os.path.join(d, 'kt', '**', '*_kernel_stats.csv'))
    for r in csv.DictReader(open(f)):
        if 'pnp::' in r['Name']:
            stats.append({'kernel': short(r['Name']), 'calls': int(r['Calls']), 'avg_us': float(r['AverageNs']) / 1e3,
                          'min_us': float(r['MinNs']) / 1e3, 'max_us': float(r['MaxNs']) / 1e3,
                          'total_ms': float(r['TotalDurationNs']) / 1e6, 'pct': float(r['Percentage'])})
    pmc = collections.defaultdict(dict)
    for sub, ctr in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
        fc = newest(os.path.join(d, sub, '**', '*_counter_collection.csv'))
        if not fc:
            continue
        acc, meta = collections.defaultdict(list), {}
        for r in csv.DictReader(open(fc)):
            if 'pnp::' in r['Kernel_Name'] and r['Counter_Name'] == ctr:
                k = short(r['Kernel_Name'])
                acc[k].append(float(r['Counter_Value']))
                meta[k] = (int(r['VGPR_Count']), int(r['LDS_Block_Size']), int(r['Grid_Size']), int(r.get('Scratch_Size', 0) or 0))
        for k, v in acc.items():
            pmc[k][ctr + '_KiB_avg'] = sum(v) / len(v)
            pmc[k][ctr + '_KiB_total'] = sum(v)
            pmc[k]['launches_' + ctr] = len(v)
            pmc[k]['vgpr'], pmc[k]['lds_bytes'], pmc[k]['grid_threads'], pmc[k]['scratch_bytes'] = meta[k]
    for k, dd in pmc.items():
        dd['read_bytes_corrected'] = 2.0 * dd.get('FETCH_SIZE_KiB_avg', 0) * 1024
        dd['write_bytes'] = dd.get('WRITE_SIZE_KiB_avg', 0) * 1024
        dd['hbm_bytes_per_launch'] = dd['read_bytes_corrected'] + dd['write_bytes']
    bench = {}
    for sub in ('kt', 'fetch'):
        for line in open(os.path.join(d, sub + '.log')):
            if line.startswith('{"metric"'):
                bench[sub] = json.loads(line)
    bk, bp = bench.get('kt', {}), bench.get('fetch', {})
    it_kt = bk.get('steps', 100) + bk.get('warmup', 10)
    it_pmc = bp.get('steps', 20) + bp.get('warmup', 2)
    generic = bk.get('config', {}).get('path') == 'generic'
    loop = [s for s in stats if (s['calls'] >= 50 if generic else any(s['kernel'].startswith(p) for p in LOOP))]
    hbm_it = sum((2.0 * pmc[s['kernel']].get('FETCH_SIZE_KiB_total', 0) + pmc[s['kernel']].get('WRITE_SIZE_KiB_total', 0)) * 1024
                 for s in loop if s['kernel'] in pmc) / it_pmc
    B = bk.get('config', {}).get('slices_per_gpu', 512)
    side = 512 if '512x512' in bk.get('config', {}).get('workload', '') else 256
    out = {'tag': tag, 'name': name,
           'command': 'bash profiles/collect.sh %s ...: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --warmup 10 '
                      '--no-cpu-baseline <args>  (+ separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes with --steps 20 --warmup 2)' % name,
           'bench_line_under_kernel_trace': bk, 'kernel_stats': stats, 'pmc': pmc,
           'per_iteration': {
               'loop_kernels': {s['kernel']: {'launches_per_iteration': round(s['calls'] / it_kt, 3), 'avg_us': s['avg_us'],
                                              'us_per_iteration': s['avg_us'] * s['calls'] / it_kt} for s in loop},
               'sum_kernel_time_us': sum(s['avg_us'] * s['calls'] / it_kt for s in loop),
               'hip_event_us_per_iteration': bk.get('hip_event_ms_per_step', 0) * 1e3,
               'hbm_bytes': hbm_it,
               'algorithmic_bytes_57N': 57 * side * side * B}}
    json.dump(out, open(os.path.join(ROOT, 'profiles', 'rocprof_%s_%s.json' % (tag, name)), 'w'), indent=1)
    with open(os.path.join(ROOT, 'profiles', 'rocprof_%s_%s_kernel_stats.csv' % (tag, name)), 'w') as fo:
        fo.write(open(f).read())
    print('== %s: %.1f it/s under the profiler, loop kernels %.1f us/iteration, %.1f MB/iteration = %.2f x 57N' % (
        name, bk.get('value', 0), out['per_iteration']['sum_kernel_time_us'], hbm_it / 1e6, hbm_it / (57.0 * side * side * B)))
    for s in stats:
        dd = pmc.get(s['kernel'], {})
        print('  %-44s calls %4d avg %9.1f us  vgpr %3s  hbm/launch %8.1f MB -> %5.2f TB/s' % (
            s['kernel'][:44], s['calls'], s['avg_us'], dd.get('vgpr', '-'), dd.get('hbm_bytes_per_launch', 0) / 1e6,
            dd.get('hbm_bytes_per_launch', 0) / (s['avg_us'] * 1e-6) / 1e12 if s['avg_us'] else 0))
    # bench.py reads this for roofline.traffic: key = path:solver:size:precision:bB
    cfg = bk.get('config', {})
    if cfg:
        solver = 'l1' if 'ADMM_L1' in cfg.get('workload', '') else 'cnc'
        key = '%s:%s:%d:%s:b%d' % (cfg.get('path'), solver, side, cfg.get('precision', 'f32'), B)
        tj = os.path.join(ROOT, 'profiles', 'traffic.json')
        cur = json.load(open(tj)) if os.path.exists(tj) else {}
        cur = {k: v for k, v in cur.items() if ':' in k}
        cur[key] = {'hbm_bytes_per_iteration': hbm_it, 'from': 'profiles/rocprof_%s_%s.json' % (tag, name)}
        json.dump(cur, open(tj, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    for nm in sys.argv[2:]:
        one(sys.argv[1], nm)
