# per-kernel breakdown of config 3 (FFDNet, 512 x 256^2) with the HIP conv backend
set -e
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/r4g/ffd
rm -rf $D; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D/kt -- python3 $R/bench_pnp.py --model ffdnet_gray --batch 512 --steps 3 --warmup 1 --cnn-backend hip > $D/kt.log 2>&1
find $D -name '*_kernel_trace.csv' -delete
python3 - <<PY
import csv, glob
for f in glob.glob('$D/kt/**/*kernel_stats.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r['TotalDurationNs']))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    for r in rows[:18]:
        print('%-100s calls %5s avg %9.1f us  total %8.2f ms %5.1f%%' % (r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, 100*float(r['TotalDurationNs'])/tot))
PY
