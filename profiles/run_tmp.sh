cd $GRAFT_REPO_ROOT
export PNP_BENCH_CACHE=/tmp/pb
timeout -k 10 300 python -m pytest tests/test_gpu_slice.py -x -q 2>&1 | tail -2
for i in 1 2 3; do echo "100 steps $(python3 bench.py --no-cpu-baseline | grep -o '"value": [0-9.]*') | 20 steps $(python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline | grep -o '"value": [0-9.]*') | l1 $(python3 bench.py --solver l1 --no-cpu-baseline | grep -o '"value": [0-9.]*')"; done
PNP_SLICE_PROF=/tmp/prof.bin timeout -k 10 200 python3 bench.py --steps 50 --warmup 0 --no-cpu-baseline > /dev/null; python3 profiles/slice_prof.py /tmp/prof.bin | grep -E "median|makespan"
