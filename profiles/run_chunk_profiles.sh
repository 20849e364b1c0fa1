#!/bin/bash
# Run ON THE GPU BOX: only the configurations whose schedule is chunked (512x512 and the double engine) + their bench lines.
cd $GRAFT_REPO_ROOT
PNP_FUSED_STREAMS=1 bash profiles/collect.sh fused512_cnc_seq --size 512 --batch 256
bash profiles/collect.sh fused512_cnc --size 512 --batch 256
PNP_FUSED_STREAMS=1 bash profiles/collect.sh fused_f64_chunk --precision f64
bash profiles/collect.sh fused_f64_default --precision f64
O=gpurun_out/bench_lines; mkdir -p $O
export PNP_BENCH_CACHE=/tmp/pnp_bench_inputs
python3 bench.py --precision f64 --steps 50 --warmup 5 --no-cpu-baseline > $O/f64_50.json 2>/dev/null
python3 bench.py --size 512 --batch 256 --no-cpu-baseline > $O/size512_100.json 2>/dev/null
python3 bench.py --size 512 --batch 256 --steps 20 --warmup 5 --no-cpu-baseline > $O/size512_driver_shape.json 2>/dev/null
for f in $O/f64_50.json $O/size512_100.json $O/size512_driver_shape.json; do echo "$(basename $f): $(grep -o '"value": [0-9.]*' $f | head -1) $(grep -o '"ms_per_step": [0-9.]*' $f) $(grep -o '"frac_measured": [0-9.a-z]*' $f)"; done
