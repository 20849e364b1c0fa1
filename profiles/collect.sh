#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:
#   bash profiles/collect.sh <name> [bench.py args...]      (environment variables such as PNP_SLICE=0 pass through)
# Writes rocprofv3 kernel-trace stats and the two PMC passes (separate runs, counters only) under
# gpurun_out/prof_<name>/{kt,fetch,write}; afterwards, in the build container:
#   python profiles/summarize.py <tag> <name> [more names...]
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; shift
D=$R/gpurun_out/prof_$NAME
rm -rf $D; mkdir -p $D
export PNP_BENCH_CACHE=/tmp/pnp_bench_inputs
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D/kt -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" > $D/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline "$@" > $D/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/write -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline "$@" > $D/write.log 2>&1
# keep only what summarize.py reads (the traces are large)
find $D -name '*_kernel_trace.csv' -delete
echo "$NAME: $(grep -h '"metric"' $D/kt.log | grep -o '"value": [0-9.]*')"
