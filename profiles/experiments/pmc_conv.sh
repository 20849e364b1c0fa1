# HBM traffic and SQ counters of k_conv3x3_c64 at [64, 64, 128, 128] (separate rocprofv3 --pmc passes, counters only)
set -e
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/r4l/pmc
rm -rf $D; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $D/g$i -- python3 $R/profiles/experiments/probe_conv.py > $D/g$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('$D/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_conv3x3_c64' in r.get('Kernel_Name', ''):
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
print('# k_conv3x3_c64 at [64, 64, 128, 128] (77.3 GFLOP, 18.87 M MFMAs, 8192 tiles on 512 persistent workgroups); per launch, sums over the 8 XCDs')
for k, (v, n) in sorted(acc.items()):
    print('%-30s %.5g  (%d launches)' % (k, v / max(n, 1), n))
f, w = acc.get('FETCH_SIZE'), acc.get('WRITE_SIZE')
if f and w:
    rb, wb = 2 * f[0] / f[1] * 1024, w[0] / w[1] * 1024
    print('# HBM-side bytes per launch: read %.1f MB (FETCH_SIZE x 2 x 1 KiB), written %.1f MB; algorithmic: 268.4 MB in (x 1.41 with the halo of 8 x 16 tiles), 268.4 MB out' % (rb / 1e6, wb / 1e6))
PY
