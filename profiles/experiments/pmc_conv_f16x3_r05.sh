# round 5: counters of k_conv3x3_c64_h3<1> in the split -> split form at [640, 64, 128, 128] (separate rocprofv3 passes, counters only)
set -e
R=$GRAFT_REPO_ROOT
N=640
D=$R/gpurun_out/pmc_h3_r05
rm -rf $D; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D/kt -- python3 $R/profiles/experiments/probe_conv_fmt.py $N 5 > $D/kt.log 2>&1
i=0
for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $D/g$i -- python3 $R/profiles/experiments/probe_conv_fmt.py $N 5 > $D/g$i.log 2>&1 || echo "pass $i failed"
done
find $D -name '*_kernel_trace.csv' -delete
cat $D/kt.log | grep -v "^$" | tail -3
python3 - <<PY
import csv, glob, collections
N = $N
for f in glob.glob('$D/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_conv3x3_c64_h3' in r['Name']:
            dur = float(r['AverageNs']); print('k_conv3x3_c64_h3<1>, split -> split, random and zero launches together, at [%d, 64, 128, 128]: avg %.1f us over %s calls' % (N, dur / 1e3, r['Calls']))
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('$D/g*/**/*counter_collection.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if 'k_conv3x3_c64_h3' in r.get('Kernel_Name', '')]
    # the probe launches 45 random-data calls, then 45 zero-data calls: keep the first 45 (random) per counter
    per = collections.defaultdict(list)
    for r in rows:
        per[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in per.items():
        half = len(v) // 2
        acc[k + ' (random)'] = [sum(v[:half]), half]
        acc[k + ' (zeros)'] = [sum(v[half:]), len(v) - half]
for k, (v, n) in sorted(acc.items()):
    print('%-44s per launch %.5g  (%d launches)' % (k, v / max(n, 1), n))
mf = N * 128 * 4 * 432
for d in ('random', 'zeros'):
    g = acc['GRBM_GUI_ACTIVE (%s)' % d]; m = acc['SQ_VALU_MFMA_BUSY_CYCLES (%s)' % d]
    cyc = g[0] / g[1] / 8
    print('# %s: GRBM_GUI_ACTIVE / 8 = %.4g cycles per launch; matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles) = %.3f (MFMAs issued: %d x 16 cycles = %.4g)' % (d, cyc, m[0] / m[1] / (1024 * cyc), mf, mf * 16.0))
PY
