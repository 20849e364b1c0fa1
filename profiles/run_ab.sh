#!/bin/bash
# A/B on ONE GPU box: the library built from build/ab/old/csrc (a copy of an earlier commit's sources) against the in-tree one.
cd $GRAFT_REPO_ROOT
export PNP_BENCH_CACHE=/tmp/pb
make -C build/ab/old/csrc -j8 OUT=../libold.so > /dev/null 2>&1
ls -la build/ab/old/libold.so
for rep in 1 2 3; do
  for args in "--precision f64 --steps 50 --warmup 5" "--size 512 --batch 256" "" ; do
    a=$(PNP_MRI_LIB=$GRAFT_REPO_ROOT/build/ab/old/libold.so PNP_SLICE=0 python3 bench.py --no-cpu-baseline $args | grep -o '"ms_per_step": [0-9.]*')
    b=$(PNP_SLICE=0 python3 bench.py --no-cpu-baseline $args | grep -o '"ms_per_step": [0-9.]*')
    echo "[$args] old $a | new $b"
  done
done
