#!/bin/bash
# Experiment helper, run ON THE GPU BOX: rebuild ONE translation unit with each set of extra -D flags, relink, bench (same box).
#   bash profiles/run_variant.sh <file.hip> "<bench args>" "<flags 1>" "<flags 2>" ...
cd $GRAFT_REPO_ROOT/pnp_admm_cnc_mri_amd/csrc
F=$1; ARGS=$2; shift 2
export PNP_BENCH_CACHE=/tmp/pb
for FLAGS in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off $FLAGS -c $F -o ${F%.hip}.o 2>&1 | grep -E "error"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpnpmri.so api.o kernels_generic.o kernels_fused256.o kernels_fused512.o kernels_slice256.o -ldl
  echo "== $F [$FLAGS] | $ARGS: $(cd ../.. && for i in 1 2 3; do python3 bench.py --no-cpu-baseline $ARGS | grep -o '"value": [0-9.]*' | cut -d' ' -f2; done | tr '\n' ' ')"
done
