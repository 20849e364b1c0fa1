#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash profiles/collect.sh [extra env]
# Writes rocprofv3 kernel-trace stats and the two PMC passes under gpurun_out/ ; afterwards run
#   python profiles/summarize.py <tag>     in the build container to condense them into profiles/.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_kt $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/prof_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_write.log 2>&1
grep -h '"metric"' $R/gpurun_out/prof_kt.log | cut -c1-200
