#!/usr/bin/env python3
"""Which schedule of the fused 256x256 loop should be the default?  Runs bench.py as FRESH processes
(as the driver does: `--steps 20 --warmup 5` and the builder's `--steps 100 --warmup 10`) for each
schedule, interleaved, R repeats each, and writes the table to gpurun_out/sched_compare.json.
Run ON THE GPU BOX:  python3 profiles/sched_compare.py [R]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = int(sys.argv[1]) if len(sys.argv) > 1 else 5
SCHEDULES = {                       # name -> (PNP_FUSED_STREAMS, PNP_FUSED_SCHED)
    'sequential_1q': ('1', '0'),
    'mixed_1q': ('1', '1'),
    'sequential_2q': ('2', '0'),
    'mixed_2q': ('2', '1'),
}
SHAPES = [(20, 5), (100, 10)]
extra = sys.argv[2:]                # e.g. --size 512 --batch 256
res = {}
for rep in range(R):
    for steps, warm in SHAPES:
        for name, (q, m) in SCHEDULES.items():
            env = dict(os.environ, PNP_FUSED_STREAMS=q, PNP_FUSED_SCHED=m, PNP_BENCH_CACHE='/tmp/pnp_bench_inputs')
            out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', str(steps), '--warmup', str(warm),
                                  '--no-cpu-baseline'] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
            j = json.loads([l for l in out.splitlines() if l.startswith('{')][-1])
            res.setdefault('%s steps=%d warmup=%d' % (name, steps, warm), []).append(
                {'value': j['value'], 'ms_per_step': j['ms_per_step'], 'hip_event_ms_per_step': j['hip_event_ms_per_step']})
            print(rep, name, steps, warm, '%.1f it/s  wall %.4f ms  events %.4f ms' % (j['value'], j['ms_per_step'], j['hip_event_ms_per_step']), flush=True)
summary = {}
for k, v in res.items():
    vals = sorted(x['value'] for x in v)
    summary[k] = {'median_it_s': vals[len(vals) // 2], 'min_it_s': vals[0], 'max_it_s': vals[-1], 'runs': v}
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump({'repeats': R, 'extra_args': extra, 'summary': summary}, open(os.path.join(ROOT, 'gpurun_out', 'sched_compare.json'), 'w'), indent=1)
for k, v in summary.items():
    print('%-40s median %.1f  min %.1f  max %.1f' % (k, v['median_it_s'], v['min_it_s'], v['max_it_s']))
