#!/bin/bash
# Experiment helper, run ON THE GPU BOX: rebuild kernels_slice256.hip with each flag set; phase clocks at batch 64 (memory
# system idle) and 512.  Ablated variants compute wrong results on purpose (timing only).  The in-tree .so is restored last.
cd $GRAFT_REPO_ROOT/pnp_admm_cnc_mri_amd/csrc
export PNP_BENCH_CACHE=/tmp/pb
cp ../libpnpmri.so /tmp/libpnpmri.keep
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off $v -c kernels_slice256.hip -o /tmp/ks_var.o 2>&1 | grep -E "error"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpnpmri.so api.o kernels_generic.o kernels_fused256.o kernels_fused512.o /tmp/ks_var.o -ldl
  echo "== variant: [$v]"
  for b in 64 512; do
    (cd ../.. && PNP_SLICE=1 PNP_SLICE_PROF=/tmp/prof.bin timeout -k 10 200 python3 bench.py --batch $b --steps 20 --warmup 0 --no-cpu-baseline | grep -o '"value": [0-9.]*' | tr '\n' ' '; echo "(batch $b)"; python3 profiles/slice_prof.py /tmp/prof.bin | grep -E "median" | grep -v "workgroup\|per-iter" | awk '{printf "   %s %s", $1, $3} END {print ""}')
  done
done
cp /tmp/libpnpmri.keep ../libpnpmri.so
