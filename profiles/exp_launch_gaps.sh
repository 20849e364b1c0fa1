cd /tmp && export TMPDIR=/tmp PNP_BENCH_CACHE=/tmp/pb
R=$GRAFT_REPO_ROOT
for b in 1 16 48; do
  rm -rf /tmp/kt_$b
  PNP_SLICE=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$b -- python3 $R/bench.py --batch $b --steps 200 --warmup 10 --no-cpu-baseline --no-f64-record > /tmp/kt_$b.log 2>&1
  echo "== batch $b: $(grep -o '"ms_per_step": [0-9.]*' /tmp/kt_$b.log | head -1)  $(grep -o '"hip_event_ms_per_step": [0-9.]*' /tmp/kt_$b.log | head -1)"
  f=$(find /tmp/kt_$b -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:4]:
    print('   %-44s calls %6s  avg %8.2f us' % (r['Name'][:44], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
