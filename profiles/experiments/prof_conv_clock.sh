# effective clock of the conv kernel on a long launch (n = 640 images of 128 x 128: ~6 ms): GRBM_GUI_ACTIVE / 8 / duration
set -e
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/r4f/clock
rm -rf $D; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D/kt -- python3 $R/profiles/experiments/probe_conv.py 640 128 128 > $D/kt.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $D/p1 -- python3 $R/profiles/experiments/probe_conv.py 640 128 128 > $D/p1.log 2>&1
find $D -name '*_kernel_trace.csv' -delete
python3 - <<PY
import csv, glob, collections
dur = None
for f in glob.glob('$D/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_conv3x3' in r['Name']:
            dur = float(r['AverageNs']); print('k_conv3x3 avg %.1f us over %s calls' % (dur / 1e3, r['Calls']))
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('$D/p1/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_conv3x3' in r.get('Kernel_Name', ''):
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print('%-28s per launch %.5g (%d)' % (k, v / n, n))
g = acc['GRBM_GUI_ACTIVE'][0] / acc['GRBM_GUI_ACTIVE'][1] / 8
print('effective clock %.3f GHz; MFMA busy / (1024 SIMDs x cycles) = %.3f' % (g / dur, acc['SQ_VALU_MFMA_BUSY_CYCLES'][0] / acc['SQ_VALU_MFMA_BUSY_CYCLES'][1] / (1024 * g)))
PY
tail -4 $D/kt.log
