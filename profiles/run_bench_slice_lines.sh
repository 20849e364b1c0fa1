#!/bin/bash
# Run ON THE GPU BOX after profiles/traffic.json was refreshed: the slice-path bench lines again (their roofline.traffic comes from it).
cd $GRAFT_REPO_ROOT
O=gpurun_out/bench_lines; mkdir -p $O
export PNP_BENCH_CACHE=/tmp/pnp_bench_inputs
python3 bench.py > $O/default_100.json 2>/dev/null
for i in 1 2 3; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/driver_shape_$i.json 2>/dev/null; done
python3 bench.py --solver l1 --no-cpu-baseline > $O/l1_100.json 2>/dev/null
for f in $O/default_100.json $O/driver_shape_?.json $O/l1_100.json; do echo "$(basename $f): $(grep -o '"value": [0-9.]*' $f | head -1) $(grep -o '"ms_per_step": [0-9.]*' $f) $(grep -o '"frac_measured": [0-9.a-z]*' $f)"; done
for m in 32 64 96 128 256; do ./profiles/micro/hbm_mix $m >> gpurun_out/hbm_mix_sizes2.jsonl || exit 1; done
