#!/bin/bash
# Run ON THE GPU BOX: everything DESIGN.md quotes for round 3 -- the -m gpu suite, the rocprofv3 collections of the
# configurations whose kernels changed this round, the bench lines, the PnP lines, a marker trace.  Afterwards, in the build
# container:   python profiles/summarize.py r03 slice_cnc slice_l1 generic_cnc
# (the two-launch kernels are unchanged since round 2: their rocprof_r02_* files stay the evidence)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/final/pytest.log
tail -4 gpurun_out/final/pytest.log
bash profiles/collect.sh slice_cnc
bash profiles/collect.sh slice_l1 --solver l1
bash profiles/collect.sh generic_cnc --generic
bash profiles/run_bench_lines.sh
# roctx ranges of the loop entry points in a marker trace (pnp_admm_cnc_run around the k_slice launch)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_markers -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_markers.log 2>&1)
find gpurun_out/prof_markers -name '*_kernel_trace.csv' -delete
find gpurun_out/prof_markers -name '*marker*' | head -3
