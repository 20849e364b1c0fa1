#!/bin/bash
# Experiment helper: A/B runs of -D variants of ONE translation unit against the in-tree library, on ONE box
# (boxes differ by ~5 %), interleaved so that drift hits every arm alike.  The in-tree .o / .so are NEVER touched:
# a variant is built as its own shared library under build/variants/ and selected with PNP_MRI_LIB (the
# developer knob of pnp_admm_cnc_mri_amd/_lib.py).
#
#   build (here or on the box; hipcc cross-compiles):   bash profiles/variants.sh build <file.hip> <name> "<-D flags>"
#   commit (build container: needs .git): the library of an earlier commit:   bash profiles/variants.sh commit <sha> <name>
#   run   (on the GPU box):                              bash profiles/variants.sh run "<bench args>" <reps> <name> [<name> ...]
#          "base" names the in-tree library.  Prints value (it/s) per arm and repetition.
#   knobs (here or on the box): the whole library with -DPNP_EXPERIMENT_KNOBS (PNP_SLICE_XOR / _QUEUES / _SEGMENT / _FLIP,
#          PNP_F512_QUEUES / PNP_F256S_QUEUES, PNP_GENERIC_STOCKHAM are compiled OUT of the product library since round 4):
#          bash profiles/variants.sh knobs   ->  build/variants/lib_knobs.so, then e.g.
#          PNP_MRI_LIB=build/variants/lib_knobs.so bash profiles/ab_env.sh "--steps 20 --warmup 5" 3 PNP_SLICE_FLIP 0 1
#   prof  (on the GPU box): phase clocks of the slice kernel at batch 64 and 512:  bash profiles/variants.sh prof <name> ...
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
C=$R/pnp_admm_cnc_mri_amd/csrc
V=$R/build/variants
mkdir -p $V
libof() { if [ "$1" = base ]; then echo $R/pnp_admm_cnc_mri_amd/libpnpmri.so; else echo $V/lib_$1.so; fi; }
case "$1" in
build)
  F=$2; NAME=$3; FLAGS=$4
  make -C $C -j4 > /dev/null
  { [ "$F" = kernels_conv_f16x3.hip ] || [ "$F" = kernels_conv_f16x3_wide.hip ] || [ "$F" = kernels_pix2x2_f16x3.hip ]; } && FLAGS="$FLAGS -fno-slp-vectorize"     # the Makefile's per-file flag
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off $FLAGS -c $C/$F -o $V/${NAME}_${F%.hip}.o
  OBJS=""
  for o in api kernels_generic kernels_fused256 kernels_fused512 kernels_slice256 kernels_conv kernels_conv_f16x3 kernels_conv_f16x3_wide kernels_pix2x2_f16x3; do
    if [ "$o.hip" = "$F" ]; then OBJS="$OBJS $V/${NAME}_$o.o"; else OBJS="$OBJS $C/$o.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/lib_$NAME.so $OBJS -ldl
  echo "built $V/lib_$NAME.so [$FLAGS]"
  ;;
knobs)
  mkdir -p $V/knobs
  OBJS=""
  for o in api kernels_generic kernels_fused256 kernels_fused512 kernels_slice256 kernels_conv kernels_conv_f16x3 kernels_conv_f16x3_wide kernels_pix2x2_f16x3; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -DPNP_EXPERIMENT_KNOBS $2 -c $C/$o.hip -o $V/knobs/$o.o &
    OBJS="$OBJS $V/knobs/$o.o"
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/lib_knobs.so $OBJS -ldl
  echo "built $V/lib_knobs.so [-DPNP_EXPERIMENT_KNOBS $2]"
  ;;
commit)
  SHA=$2; NAME=$3
  rm -rf $R/build/ab/$NAME && mkdir -p $R/build/ab/$NAME
  git -C $R archive $SHA pnp_admm_cnc_mri_amd/csrc include | tar -x -C $R/build/ab/$NAME
  make -C $R/build/ab/$NAME/pnp_admm_cnc_mri_amd/csrc -j4 OUT=$V/lib_$NAME.so > /dev/null
  echo "built $V/lib_$NAME.so from $SHA"
  ;;
run)
  ARGS=$2; REPS=$3; shift 3
  export PNP_BENCH_CACHE=/tmp/pb
  cd $R
  for rep in $(seq 1 $REPS); do
    for n in "$@"; do
      v=$(PNP_MRI_LIB=$(libof $n) timeout -k 10 300 python3 bench.py --no-cpu-baseline $ARGS | grep -o '"value": [0-9.]*' | cut -d' ' -f2)
      echo "[$ARGS] rep $rep $n $v"
    done
  done
  ;;
prof)
  shift
  export PNP_BENCH_CACHE=/tmp/pb
  cd $R
  for n in "$@"; do
    for b in 64 512; do
      PNP_MRI_LIB=$(libof $n) PNP_SLICE=1 PNP_SLICE_PROF=/tmp/prof.bin timeout -k 10 300 python3 bench.py --batch $b --steps 20 --warmup 0 --no-cpu-baseline > /dev/null
      echo "== $n batch $b: $(python3 profiles/slice_prof.py /tmp/prof.bin | grep -E 'median' | grep -v 'workgroup\|per-iter' | awk '{printf "%s %s  ", $1, $3}')"
    done
  done
  ;;
*) echo "usage: see the header of $0"; exit 2;;
esac
