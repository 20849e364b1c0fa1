# round 6: counters of the 64 -> 64 conv3x3 layer in the split -> split form at [640, 64, 128, 128], NARROW kernel (k_conv3x3_c64_h3<1>, PNP_CONV_WIDE=0)
# and WIDE kernel (k_conv3x3_h3w, PNP_CONV_WIDE=1): separate rocprofv3 passes, counters only; usage (GPU box): bash profiles/experiments/pmc_conv_f16x3_r06.sh [tag]
set -e
R=$GRAFT_REPO_ROOT
N=640
TAG=${1:-r06}
D=$R/gpurun_out/pmc_h3_$TAG
rm -rf $D; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
for V in 0 1; do
  export PNP_CONV_WIDE=$V
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/v$V/kt -- python3 $R/profiles/experiments/probe_conv_fmt.py $N 5 > $D/v$V.kt.log 2>&1
  i=0
  for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --output-format csv -d $D/v$V/g$i -- python3 $R/profiles/experiments/probe_conv_fmt.py $N 5 > $D/v$V.g$i.log 2>&1 || echo "variant $V pass $i failed"
  done
  grep -v "^$" $D/v$V.kt.log | tail -2
done
find $D -name '*_kernel_trace.csv' -delete
python3 - <<PY
import csv, glob, collections
N = $N
for V, kn, label in ((0, 'k_conv3x3_c64_h3', 'NARROW k_conv3x3_c64_h3<1>'), (1, 'k_conv3x3_h3w', 'WIDE k_conv3x3_h3w')):
    for f in glob.glob('$D/v%d/kt/**/*kernel_stats.csv' % V, recursive=True):
        for r in csv.DictReader(open(f)):
            if kn in r['Name']:
                print('%s, split -> split, random and zero launches together, at [%d, 64, 128, 128]: avg %.1f us over %s calls' % (label, N, float(r['AverageNs']) / 1e3, r['Calls']))
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob('$D/v%d/g*/**/*counter_collection.csv' % V, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if kn in r.get('Kernel_Name', '')]
        per = collections.defaultdict(list)
        for r in rows:
            per[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in per.items():               # 45 random-data launches, then 45 zero-data ones
            half = len(v) // 2
            acc[k + ' (random)'] = [sum(v[:half]), half]
            acc[k + ' (zeros)'] = [sum(v[half:]), len(v) - half]
    for k, (v, n) in sorted(acc.items()):
        print('  %-44s per launch %.5g  (%d launches)' % (k, v / max(n, 1), n))
    mf = N * 128 * 4 * 432
    for d in ('random', 'zeros'):
        g = acc['GRBM_GUI_ACTIVE (%s)' % d]; m = acc['SQ_VALU_MFMA_BUSY_CYCLES (%s)' % d]
        if not g[1] or not m[1]:
            continue
        cyc = g[0] / g[1] / 8
        print('# %s, %s: GRBM_GUI_ACTIVE / 8 = %.4g cycles per launch; matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles) = %.3f (MFMAs issued: %d x 16 cycles = %.4g)' % (label, d, cyc, m[0] / m[1] / (1024 * cyc), mf, mf * 16.0))
PY
