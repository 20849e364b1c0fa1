cd $GRAFT_REPO_ROOT
export PNP_BENCH_CACHE=/tmp/pb
for B in 64 128 192 256 384 512 768 1024; do
  for m in 0 1; do
    echo "B=$B slice=$m $(PNP_SLICE=$m timeout -k 10 300 python3 bench.py --batch $B --steps 100 --warmup 10 --no-cpu-baseline | grep -o '"ms_per_step": [0-9.]*')"
  done
done
echo driver-shaped:
for i in 1 2 3; do for m in 0 1; do echo "slice=$m $(PNP_SLICE=$m timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline | grep -o '"value": [0-9.]*')"; done; done
