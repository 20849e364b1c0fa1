#!/bin/bash
# Run ON THE GPU BOX: the bench lines quoted in DESIGN.md / README.md, written to gpurun_out/bench_lines/.
cd $GRAFT_REPO_ROOT
O=gpurun_out/bench_lines; mkdir -p $O
export PNP_BENCH_CACHE=/tmp/pnp_bench_inputs
python3 bench.py > $O/default_100.json 2>/dev/null                                         # full line incl. CPU baselines, parity and f64 records
python3 bench.py --steps 20 --warmup 5 > $O/driver_shape_full.json 2>/dev/null                 # the driver's command line
for i in 1 2 3; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/driver_shape_$i.json 2>/dev/null; done
PNP_SLICE=0 python3 bench.py --no-cpu-baseline > $O/fused_100.json 2>/dev/null
python3 bench.py --solver l1 --no-cpu-baseline > $O/l1_100.json 2>/dev/null
python3 bench.py --precision f64 --steps 50 --warmup 5 --no-cpu-baseline > $O/f64_50.json 2>/dev/null
python3 bench.py --size 512 --batch 256 --no-cpu-baseline > $O/size512_100.json 2>/dev/null
python3 bench.py --size 512 --batch 256 --steps 20 --warmup 5 --no-cpu-baseline > $O/size512_driver_shape.json 2>/dev/null
python3 bench.py --generic --no-cpu-baseline > $O/generic_100.json 2>/dev/null
# (the round-2 Stockham column kernel is an experiment-build knob since round 4: bash profiles/variants.sh knobs; PNP_MRI_LIB=build/variants/lib_knobs.so PNP_GENERIC_STOCKHAM=1 ...)
python3 bench.py --gpus 2 --rehearse-gloo --steps 20 --warmup 5 > $O/gpus2_rehearsal.json 2>/dev/null
PNP_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/rccl_one_rank.json 2>/dev/null
python3 bench.py --batch 1024 --no-cpu-baseline > $O/batch1024_100.json 2>/dev/null
python3 - $O <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(sys.argv[1] + '/*.json')):
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l); r = d.get('roofline', {})
            su = d.get('sustained') or {}
            print('%s: value %.1f  ms_per_step %.5f  frac %s  | sustained %s it/s frac %s' % (os.path.basename(f), d['value'], d['ms_per_step'], r.get('frac'), su.get('value'), su.get('frac')))
PY
for be in torch hip; do
python3 bench_pnp.py --model ffdnet_gray --batch 512 --steps 6 --warmup 2 --cnn-backend $be > $O/pnp_ffdnet_$be.json 2>/dev/null; tail -1 $O/pnp_ffdnet_$be.json | cut -c1-300
python3 bench_pnp.py --model drunet_gray --batch 512 --steps 2 --warmup 1 --cnn-backend $be > $O/pnp_drunet_$be.json 2>/dev/null; tail -1 $O/pnp_drunet_$be.json | cut -c1-300
python3 bench_pnp.py --model drunet_gray --size 512 --batch 64 --cnn-batch 16 --steps 2 --warmup 1 --cnn-backend $be > $O/pnp_drunet512_$be.json 2>/dev/null; tail -1 $O/pnp_drunet512_$be.json | cut -c1-300
python3 bench_pnp.py --model dncnn_15 --batch 512 --steps 3 --warmup 1 --cnn-backend $be > $O/pnp_dncnn15_$be.json 2>/dev/null; tail -1 $O/pnp_dncnn15_$be.json | cut -c1-300
done
python3 bench_pnp.py --model ffdnet_gray --batch 64 --steps 3 --warmup 1 --gpus 2 --rehearse-gloo > $O/pnp_ffdnet_gpus2_rehearsal.json 2>/dev/null; tail -1 $O/pnp_ffdnet_gpus2_rehearsal.json | cut -c1-200
# the split-half (f16x3) convolution backend: every model family
for m in "ffdnet_gray 512 6 2" "dncnn_15 512 3 1" "ircnn_gray 512 3 1" "drunet_gray 512 2 1"; do set -- $m
python3 bench_pnp.py --model $1 --batch $2 --steps $3 --warmup $4 --cnn-backend hip_f16x3 > $O/pnp_$1_f16x3.json 2>/dev/null; tail -1 $O/pnp_$1_f16x3.json | cut -c1-300
done
python3 bench_pnp.py --model drunet_gray --size 512 --batch 64 --cnn-batch 16 --steps 2 --warmup 1 --cnn-backend hip_f16x3 > $O/pnp_drunet512_f16x3.json 2>/dev/null; tail -1 $O/pnp_drunet512_f16x3.json | cut -c1-300
