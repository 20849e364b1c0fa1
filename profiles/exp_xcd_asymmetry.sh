#!/bin/bash
# Run ON THE GPU BOX (needs build/variants/lib_prof.so = -DSLICE_PROF, profiles/variants.sh build): where do the slow
# workgroups of the slice-resident kernel come from?  Phase clocks by blockIdx % 8 (= XCD under round-robin dispatch) with
# the slice -> workgroup map unchanged, with neighbouring slices swapped (PNP_SLICE_XOR=1) and with slices swapped inside
# their parity class (PNP_SLICE_XOR=8).  Result (round 3, gpurun_out/exp_xcd.txt -> DESIGN.md 4.1): odd-numbered SLICES run
# their memory phases 15-20 % slower than even ones whichever XCD hosts them; profiles/micro/slice_stride.hip reproduces
# it with a plain streaming kernel, with any padding of the 256 KiB slice stride below another 256 KiB.
export PNP_BENCH_CACHE=/tmp/pb
R=$GRAFT_REPO_ROOT
run() { echo "=== $1"; env $1 PNP_MRI_LIB=$R/build/variants/lib_prof.so PNP_SLICE_PROF=/tmp/prof.bin python3 $R/bench.py --steps 50 --warmup 0 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*'; python3 $R/profiles/slice_prof.py /tmp/prof.bin | grep -A9 "duration by blockIdx\|first-round medians" | head -14; }
run "PNP_X=0"
run "PNP_SLICE_XOR=1"
run "PNP_SLICE_XOR=8"
for v in "PNP_X=0" "PNP_SLICE_XOR=1"; do echo "$v: $(env $v python3 $R/bench.py --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*') $(env $v python3 $R/bench.py --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*')"; done
