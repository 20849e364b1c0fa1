#!/bin/bash
# Run ON THE GPU BOX: slice-resident vs two-launch path by batch size (same box), for the selection rule in api.hip
cd $GRAFT_REPO_ROOT
export PNP_BENCH_CACHE=/tmp/pb
for B in 96 128 160 192 224 256 288 320 352 384 448 512 640 768; do
  a=$(PNP_SLICE=0 python3 bench.py --batch $B --steps 100 --warmup 10 --no-cpu-baseline | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2)
  b=$(PNP_SLICE=1 python3 bench.py --batch $B --steps 100 --warmup 10 --no-cpu-baseline | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2)
  echo "B=$B fused $a slice $b"
done
