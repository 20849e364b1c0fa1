#!/bin/bash
# usage: ab_env.sh "<bench args>" reps VAR val1 val2 ...
ARGS=$1; REPS=$2; VAR=$3; shift 3
export PNP_BENCH_CACHE=/tmp/pb
for rep in $(seq 1 $REPS); do for v in "$@"; do
  r=$(env $VAR=$v timeout -k 10 300 python3 bench.py --no-cpu-baseline $ARGS 2>/dev/null | grep -o '"value": [0-9.]*' | cut -d' ' -f2)
  echo "[$ARGS] rep $rep $VAR=$v $r"
done; done
