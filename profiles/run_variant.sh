#!/bin/bash
# Experiment helper, run ON THE GPU BOX: rebuild ONE translation unit with extra -D flags, relink, bench.
#   bash profiles/run_variant.sh <file.hip> "<flags>" [bench.py args...]
cd $GRAFT_REPO_ROOT/pnp_admm_cnc_mri_amd/csrc
F=$1; FLAGS=$2; shift 2
export PNP_BENCH_CACHE=/tmp/pb
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off $FLAGS -c $F -o ${F%.hip}.o 2>&1 | grep -E "error"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpnpmri.so api.o kernels_generic.o kernels_fused256.o kernels_fused512.o kernels_slice256.o -ldl
echo "== $F $FLAGS | $@"
cd ../.. && for i in 1 2 3; do timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" | grep -o '"ms_per_step": [0-9.]*'; done
