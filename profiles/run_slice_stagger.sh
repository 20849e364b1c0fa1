cd $GRAFT_REPO_ROOT
export PNP_BENCH_CACHE=/tmp/pb
for st in 0 10 20 30; do
  echo "== stagger $st us"
  for i in 1 2; do PNP_SLICE_STAGGER_US=$st PNP_SLICE=1 timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline | grep -o '"value": [0-9.]*'; done
  PNP_SLICE_STAGGER_US=$st PNP_SLICE=1 PNP_SLICE_PROF=/tmp/prof.bin timeout -k 10 200 python3 bench.py --steps 30 --warmup 0 --no-cpu-baseline > /dev/null; python3 profiles/slice_prof.py /tmp/prof.bin | grep -E "median"
done
