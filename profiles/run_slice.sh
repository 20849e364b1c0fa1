cd $GRAFT_REPO_ROOT
export PNP_BENCH_CACHE=/tmp/pb
timeout -k 10 300 python -m pytest tests/test_gpu_slice.py -x -q 2>&1 | tail -3
for i in 1 2; do
PNP_SLICE=1 timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline | grep -o '"value": [0-9.]*'
done
PNP_SLICE=1 timeout -k 10 200 python3 bench.py --solver l1 --steps 100 --warmup 10 --no-cpu-baseline | grep -o '"value": [0-9.]*'
PNP_SLICE=1 PNP_SLICE_PROF=/tmp/prof.bin timeout -k 10 200 python3 bench.py --steps 20 --warmup 0 --no-cpu-baseline | grep -o '"value": [0-9.]*'
python3 profiles/slice_prof.py /tmp/prof.bin
