# k_conv3x3_c64_h3 (f16x3) at [N, 64, 128, 128]: kernel time, effective clock, matrix-pipe occupancy, LDS conflicts, HBM-side bytes
# (separate rocprofv3 passes: kernel trace; counters only).   usage (GPU box): bash profiles/experiments/pmc_conv_f16x3.sh [N=640]
set -e
R=$GRAFT_REPO_ROOT
N=${1:-640}
D=$R/gpurun_out/pmc_h3_$N
rm -rf $D; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D/kt -- python3 $R/profiles/experiments/probe_conv.py $N 128 128 1 f16x3 > $D/kt.log 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $D/g$i -- python3 $R/profiles/experiments/probe_conv.py $N 128 128 1 f16x3 > $D/g$i.log 2>&1 || echo "pass $i failed"
done
find $D -name '*_kernel_trace.csv' -delete
python3 - <<PY
import csv, glob, collections
N = $N
dur = None
for f in glob.glob('$D/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_conv3x3_c64_h3' in r['Name']:
            dur = float(r['AverageNs']); print('k_conv3x3_c64_h3<1> at [%d, 64, 128, 128]: avg %.1f us over %s calls = %.1f TFLOP/s of float32-equivalent arithmetic' % (N, dur / 1e3, r['Calls'], 2.0 * N * 16384 * 64 * 64 * 9 / dur / 1e3))
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('$D/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_conv3x3_c64_h3' in r.get('Kernel_Name', ''):
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print('%-30s per launch %.5g  (%d launches)' % (k, v / max(n, 1), n))
per = lambda k: acc[k][0] / acc[k][1]
g = per('GRBM_GUI_ACTIVE') / 8
mf = N * 128 * 4 * 432                      # tiles x waves x v_mfma_f32_16x16x32_f16 per wave-tile (9 taps x 48)
print('# effective clock %.3f GHz (GRBM_GUI_ACTIVE / 8 / duration); v_mfma_f32_16x16x32_f16 issued: %d = %.4g busy cycles at 16 each; counter / (1024 SIMDs x cycles) = %.3f'
      % (g / dur, mf, 16.0 * mf, per('SQ_VALU_MFMA_BUSY_CYCLES') / (1024 * g)))
rb, wb = 2 * per('FETCH_SIZE') * 1024, per('WRITE_SIZE') * 1024
alg = N * 16384 * 256
print('# HBM-side bytes per launch: read %.1f MB (FETCH_SIZE x 2 x 1 KiB), written %.1f MB; algorithmic %.1f MB in + %.1f MB out; (read + written) / duration = %.2f TB/s'
      % (rb / 1e6, wb / 1e6, alg / 1e6, alg / 1e6, (rb + wb) / dur / 1e3))
print('# LDS bank conflict cycles / LDS active cycles = %.3f' % (per('SQ_LDS_BANK_CONFLICT') / per('SQ_LDS_IDX_ACTIVE')))
PY
