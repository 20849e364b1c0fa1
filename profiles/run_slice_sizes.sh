#!/bin/bash
# Run ON THE GPU BOX: slice-resident vs two-launch path by batch size (same box), for the selection rule in api.hip
cd $GRAFT_REPO_ROOT
export PNP_BENCH_CACHE=/tmp/pb
for B in ${SIZES:-64 80 96 112 128 192 256 272 288 320 384 512 640}; do
  a=$(PNP_SLICE=0 python3 bench.py --batch $B --steps 100 --warmup 10 --no-cpu-baseline | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2)
  b=$(PNP_SLICE=1 python3 bench.py --batch $B --steps 100 --warmup 10 --no-cpu-baseline | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2)
  echo "B=$B fused $a slice $b"
done
