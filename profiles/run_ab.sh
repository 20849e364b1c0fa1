#!/bin/bash
# A/B on ONE GPU box (boxes differ by ~5 %): the library built from build/ab/old/csrc (a copy of an earlier commit's
# sources, see the command that filled it) against the in-tree one.   bash profiles/run_ab.sh "<bench args>" ...
cd $GRAFT_REPO_ROOT
export PNP_BENCH_CACHE=/tmp/pb
make -C build/ab/old/csrc -j8 OUT=../libold.so > /dev/null 2>&1
for rep in 1 2 3; do
  for args in "$@"; do
    a=$(PNP_MRI_LIB=$GRAFT_REPO_ROOT/build/ab/old/libold.so python3 bench.py --no-cpu-baseline $args | grep -o '"value": [0-9.]*')
    b=$(python3 bench.py --no-cpu-baseline $args | grep -o '"value": [0-9.]*')
    echo "[$args] old $a | new $b"
  done
done
