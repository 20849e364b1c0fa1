#!/bin/bash
# Experiment helper, run ON THE GPU BOX: rebuild kernels_slice256.hip with each flag set, bench + phase clocks.
cd $GRAFT_REPO_ROOT/pnp_admm_cnc_mri_amd/csrc
export PNP_BENCH_CACHE=/tmp/pb
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off $v -c kernels_slice256.hip -o kernels_slice256.o 2>&1 | grep -E "error"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpnpmri.so api.o kernels_generic.o kernels_fused256.o kernels_fused512.o kernels_slice256.o -ldl
  echo "== variant: $v"
  (cd ../.. && for i in 1 2; do timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline | grep -o '"value": [0-9.]*'; done; PNP_SLICE_PROF=/tmp/prof.bin timeout -k 10 200 python3 bench.py --steps 20 --warmup 0 --no-cpu-baseline > /dev/null; python3 profiles/slice_prof.py /tmp/prof.bin | grep -E "median")
done
