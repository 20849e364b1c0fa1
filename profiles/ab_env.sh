#!/bin/bash
# A/B of an environment knob on ONE box, interleaved:   bash profiles/ab_env.sh "<bench args>" <reps> <VAR> <value> [<value> ...]
# (a value may carry further assignments: "4 PNP_SLICE_YH_PAD_KB=4")
# Experiment knobs (PNP_SLICE_FLIP, PNP_SLICE_XOR, ...) exist only in the -DPNP_EXPERIMENT_KNOBS build:
#   bash profiles/variants.sh knobs && export PNP_MRI_LIB=$PWD/build/variants/lib_knobs.so
ARGS=$1; REPS=$2; VAR=$3; shift 3
export PNP_BENCH_CACHE=/tmp/pb
for rep in $(seq 1 $REPS); do for v in "$@"; do
  r=$(env $VAR=$v timeout -k 10 300 python3 bench.py --no-cpu-baseline $ARGS 2>/dev/null | grep -o '"value": [0-9.]*' | cut -d' ' -f2)
  echo "[$ARGS] rep $rep $VAR=$v $r"
done; done
