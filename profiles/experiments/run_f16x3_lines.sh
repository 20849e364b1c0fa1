cd $GRAFT_REPO_ROOT
O=gpurun_out/bench_lines; mkdir -p $O
for m in "ffdnet_gray 512 6 2" "dncnn_15 512 3 1" "ircnn_gray 512 3 1" "drunet_gray 512 2 1"; do set -- $m
python3 bench_pnp.py --model $1 --batch $2 --steps $3 --warmup $4 --cnn-backend hip_f16x3 > $O/pnp_$1_f16x3.json 2>/dev/null; python3 -c "
import json;j=json.load(open('$O/pnp_$1_f16x3.json'));print('$1 f16x3', round(j['value'],3), round(j['ms_per_step'],1), round(j['denoiser']['roofline']['frac'],3))"
done
python3 bench_pnp.py --model drunet_gray --size 512 --batch 64 --cnn-batch 16 --steps 2 --warmup 1 --cnn-backend hip_f16x3 > $O/pnp_drunet512_f16x3.json 2>/dev/null; python3 -c "
import json;j=json.load(open('$O/pnp_drunet512_f16x3.json'));print('drunet512 f16x3', round(j['value'],4), round(j['ms_per_step'],1), round(j['denoiser']['roofline']['frac'],3))"
python3 bench_pnp.py --model ffdnet_gray --batch 512 --steps 6 --warmup 2 --cnn-backend hip_f16x3 --cnn-batch 512 > $O/pnp_ffdnet_gray_f16x3_cnn512.json 2>/dev/null; python3 -c "
import json;j=json.load(open('$O/pnp_ffdnet_gray_f16x3_cnn512.json'));print('ffdnet f16x3 cnn-batch 512', round(j['value'],3), round(j['ms_per_step'],1), round(j['denoiser']['roofline']['frac'],3))"
