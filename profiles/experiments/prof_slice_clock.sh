# effective clock of the product slice kernel over launches of 20 / 100 / 1000 iterations: GRBM_GUI_ACTIVE / 8 / duration
# (MI355X_MICROARCH.md, DVFS give-back: within 3 % of the in-kernel clock on dispatches of 10 ms or more)
set -e
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/r4j/clock
rm -rf $D; mkdir -p $D
export PNP_BENCH_CACHE=/tmp/pb
cd /tmp && export TMPDIR=/tmp
for K in 20 100 1000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/kt$K -- python3 $R/bench.py --steps $K --warmup 5 --sustain-s 0 --no-cpu-baseline > $D/kt$K.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $D/p$K -- python3 $R/bench.py --steps $K --warmup 5 --sustain-s 0 --no-cpu-baseline > $D/p$K.log 2>&1
done
find $D -name '*_kernel_trace.csv' -delete
python3 - <<PY
import csv, glob
for K in (20, 100, 1000):
    dur = None
    for f in glob.glob('$D/kt%d/**/*kernel_stats.csv' % K, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_slice' in r['Name']:
                dur = float(r['MaxNs'])           # the K-iteration launch (the warm-up launch is the shorter one)
    g = []
    for f in glob.glob('$D/p%d/**/*counter_collection.csv' % K, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_slice' in r.get('Kernel_Name', '') and r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
                g.append(float(r['Counter_Value']))
    gmax = max(g)
    print('K = %4d: launch %.1f us = %.2f us per iteration; GRBM_GUI_ACTIVE %.4g -> effective clock %.3f GHz' % (K, dur / 1e3, dur / 1e3 / K, gmax, gmax / 8 / dur))
PY
