export PNP_BENCH_CACHE=/tmp/pb
R=$GRAFT_REPO_ROOT
run() { echo "=== $1"; env $1 PNP_MRI_LIB=$R/build/variants/lib_prof.so PNP_SLICE_PROF=/tmp/prof.bin python3 $R/bench.py --steps 50 --warmup 0 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*'; python3 $R/profiles/slice_prof.py /tmp/prof.bin | grep -A9 "duration by blockIdx\|first-round medians" | head -14; }
run "PNP_X=0"
run "PNP_SLICE_XOR=1"
run "PNP_SLICE_XOR=8"
run "PNP_STATE_SKEW_KB=3"
run "PNP_STATE_SKEW_KB=67"
for v in "PNP_X=0" "PNP_SLICE_XOR=1" "PNP_STATE_SKEW_KB=3" "PNP_STATE_SKEW_KB=67" "PNP_STATE_SKEW_KB=1027"; do echo "$v: $(env $v python3 $R/bench.py --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*') $(env $v python3 $R/bench.py --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*')"; done
