set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b; mkdir -p $O
export PNP_BENCH_CACHE=/tmp/pnp_bench_inputs
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
for i in 1 2 3; do
  timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/b_old_$i.log 2>&1
  PNP_FUSED_COLS=2 timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/b_split_$i.log 2>&1
  PNP_FUSED_COLS=2 PNP_FUSED_STREAMS=1 timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/b_split1q_$i.log 2>&1
done
grep -h -o '"value": [0-9.]*' $O/b_old_*.log $O/b_split_*.log $O/b_split1q_*.log
timeout -k 10 300 python3 bench.py --precision f64 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_f64_fused.log 2>&1
tail -1 $O/bench_f64_fused.log | cut -c1-300
PNP_FUSED_STREAMS=1 timeout -k 10 300 python3 bench.py --precision f64 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_f64_fused_1q.log 2>&1
tail -1 $O/bench_f64_fused_1q.log | cut -c1-300
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
PNP_FUSED_COLS=2 PNP_FUSED_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt_split -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $R/$O/kt_split.log 2>&1
PNP_FUSED_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt_f64 -- python3 $R/bench.py --precision f64 --steps 30 --warmup 3 --no-cpu-baseline > $R/$O/kt_f64.log 2>&1
cd $R
for d in kt_split kt_f64; do f=$(ls $O/$d/*/*_kernel_stats.csv | head -1); head -6 $f | cut -c1-160; done
