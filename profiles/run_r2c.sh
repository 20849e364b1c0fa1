set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
export PNP_BENCH_CACHE=/tmp/pnp_bench_inputs
timeout -k 10 600 python -m pytest tests/test_gpu_slice.py tests/test_gpu_f64.py -x -q > $O/pytest_slice.log 2>&1; echo "pytest rc $?" >> $O/pytest_slice.log
tail -15 $O/pytest_slice.log
for i in 1 2; do
  PNP_SLICE=1 timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/b_slice_$i.log 2>&1
  PNP_SLICE=1 timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b_slice20_$i.log 2>&1
  PNP_SLICE=1 timeout -k 10 200 python3 bench.py --solver l1 --steps 100 --warmup 10 --no-cpu-baseline > $O/b_slice_l1_$i.log 2>&1
  timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/b_fused_$i.log 2>&1
done
grep -h -o '"value": [0-9.]*' $O/b_slice_?.log $O/b_slice20_?.log $O/b_slice_l1_?.log $O/b_fused_?.log
