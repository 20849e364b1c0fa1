# per-kernel breakdown of a PnP bench line:  bash profiles/experiments/prof_pnp_kernels.sh <model> <backend> <steps> [extra bench_pnp args]
set -e
R=$GRAFT_REPO_ROOT
M=$1; B=$2; K=$3; shift 3
D=$R/gpurun_out/pnpk/${M}_$B
rm -rf $D; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D/kt -- python3 $R/bench_pnp.py --model $M --batch 512 --steps $K --warmup 1 --cnn-backend $B "$@" > $D/kt.log 2>&1
find $D -name '*_kernel_trace.csv' -delete
python3 - <<PY
import csv, glob
for f in glob.glob('$D/kt/**/*kernel_stats.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r['TotalDurationNs']))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print('$M $B: %.1f ms of kernels' % (tot / 1e6))
    for r in rows[:22]:
        print('%-110s calls %5s avg %9.1f us  total %8.2f ms %5.1f%%' % (r['Name'][:110], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, 100*float(r['TotalDurationNs'])/tot))
PY
