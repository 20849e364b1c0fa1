#!/bin/bash
# round 6, final code (GPU box, from the repo root): the driver's bench command with every sub-record, rocprofv3 kernel stats + PMC passes of the
# headline configuration, per-kernel breakdowns of the two PnP children.   bash profiles/run_r06_final.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_final
mkdir -p $O
cd $R
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $O/driver_shape_full.json 2> $O/driver_shape_full.err; echo "bench rc $?"
timeout -k 10 300 bash profiles/collect.sh r06_slice_cnc --sustain-s 0 --no-l1-record > $O/collect.log 2>&1; tail -1 $O/collect.log
timeout -k 10 200 bash profiles/experiments/prof_pnp_kernels.sh ffdnet_gray hip_f16x3 4 --no-parity > $O/rocprof_r06_pnp_ffdnet_f16x3_kernels.txt 2>&1
timeout -k 10 300 bash profiles/experiments/prof_pnp_kernels.sh drunet_gray hip_f16x3 2 --no-parity --mask Q_Cartesian30 > $O/rocprof_r06_pnp_drunet_f16x3_kernels.txt 2>&1
find $R/gpurun_out/pnpk -name '*kernel_stats.csv' | while read f; do cp $f $O/$(echo $f | sed 's#.*/pnpk/\([a-z_0-9]*\)/.*#rocprof_r06_pnp_\1_kernel_stats.csv#'); done
head -4 $O/rocprof_r06_pnp_ffdnet_f16x3_kernels.txt; head -4 $O/rocprof_r06_pnp_drunet_f16x3_kernels.txt
python3 -c "
import json
j=json.loads(open('$O/driver_shape_full.json').read().strip().splitlines()[-1])
print('value', j['value'], 'frac', j['roofline']['frac'], 'frac_of_calibration', j['roofline'].get('frac_of_calibration'), 'calib', j['roofline'].get('calibration'))
print('device', j['config'].get('device'))
print('sustained', j['sustained']['value'], 'l1', j['l1']['value'] if j.get('l1') else None, 'f64', j['f64']['value'] if j.get('f64') else None)
print('latency', json.dumps(j.get('latency')))
p=j.get('pnp',{})
for k in ('torch','hip_f16x3','config4_shard_drunet_hip_f16x3'):
    r=p.get(k,{}); print(k, r.get('value'), r.get('denoiser_roofline',{}).get('frac'), r.get('sustained'), r.get('parity',{}).get('rel_l2_vs_oracle') if r.get('parity') else None, r.get('error'))
print('cpu', j['cpu_baseline']['value'], j.get('cpu_baseline_all_cores',{}).get('value'))
"
