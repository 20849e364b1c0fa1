set -e
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/r4d/prof
rm -rf $D; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D/kt -- python3 $R/profiles/experiments/probe_conv.py > $D/kt.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $D/p1 -- python3 $R/profiles/experiments/probe_conv.py > $D/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $D/p2 -- python3 $R/profiles/experiments/probe_conv.py > $D/p2.log 2>&1
find $D -name '*_kernel_trace.csv' -delete
python3 - <<PY
import csv, glob, collections
for f in glob.glob('$D/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv' in r['Name'] or 'Conv' in r['Name'] or 'igemm' in r['Name'].lower() or 'miopen' in r['Name'].lower():
            print('%-90s calls %4s avg %9.1f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('$D/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_conv3x3' in r.get('Kernel_Name', ''):
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print('%-32s per launch %.4g  (%d records)' % (k, v / max(n, 1), n))
PY
