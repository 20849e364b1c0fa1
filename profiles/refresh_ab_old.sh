#!/bin/bash
# In the build container: fill build/ab/old with the sources of a commit (default HEAD) and build libold.so, the "old" side of
# profiles/run_ab.sh.   bash profiles/refresh_ab_old.sh [commit]
set -e
cd "$(dirname "$0")/.."
C=${1:-HEAD}
rm -rf build/ab && mkdir -p build/ab/old/csrc build/ab/include build/ab/tmp
git archive $C pnp_admm_cnc_mri_amd/csrc include | tar -x -C build/ab/tmp
cp -r build/ab/tmp/pnp_admm_cnc_mri_amd/csrc/* build/ab/old/csrc/ && cp build/ab/tmp/include/* build/ab/include/ && rm -rf build/ab/tmp
make -C build/ab/old/csrc -j8 OUT=../libold.so > /dev/null
ls -la build/ab/old/libold.so
