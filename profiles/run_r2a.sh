set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
timeout -k 10 600 python3 profiles/sched_compare.py 5 > gpurun_out/r2a/sched.log 2>&1 && cp gpurun_out/sched_compare.json gpurun_out/r2a/sched_compare_256.json
tail -9 gpurun_out/r2a/sched.log
timeout -k 10 300 python3 bench.py --precision f64 --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r2a/bench_f64_generic.log 2>&1
tail -1 gpurun_out/r2a/bench_f64_generic.log | cut -c1-400
timeout -k 10 300 python3 bench.py > gpurun_out/r2a/bench_default.log 2>&1
tail -1 gpurun_out/r2a/bench_default.log | cut -c1-1500
