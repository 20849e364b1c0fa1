#!/usr/bin/env python3
"""Auxiliary benchmark (NOT the driver's headline bench.py): the plug-and-play configurations of BASELINE.json.

    config 3   python bench_pnp.py --model ffdnet_gray --batch 512                      (FFDNet, Q_Radial30, 1 GPU)
    config 4   python bench_pnp.py --model drunet_gray --batch 512 --gpus 8             (DRUNet, Q_Cartesian30, 8 x 512 slices)
    config 5   python bench_pnp.py --model drunet_gray --size 512 --batch 256 --gpus 8  (512 x 512, mask bank of 3, 8 x 256 slices)

One step = one PNP_ADMM_CNC_D iteration (S6:266-308) over the slices a GPU holds: x-update in k-space (HIP, pnp_dc_step),
s = D(z), the CNC combination (HIP), z = D(t), dual update + clamps (HIP).  Slices shard over the GPUs exactly as in
bench.py (contiguous blocks, no data-path collective, one RCCL gather of x at the end, timed apart); `--gpus N` starts its
own rank processes through bench.launch_ranks.  Seeded synthetic weights and inputs (no network).

SURVEY.md 8(d) asks, for the PnP configs, for the FFT + prox part's own figure and, separately, the denoiser's time share and
FLOP/s: `fft_prox` carries time, algorithmic bytes (57 N per slice-iteration) and the HBM fraction; `denoiser` carries time
share, FLOP per call (2 x the MACs of every convolution actually executed, counted by hooks) and FLOP/s against the fp32
matrix-core peak of MI355X (157.3 TFLOP/s: MIOpen's fp32 convolutions are what the north star keeps there).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32_MATRIX_PEAK_GFLOPS = 157300.0            # MI355X_MICROARCH.md: fp32 MFMA = fp32 vector peak
F16_MATRIX_PEAK_GFLOPS = 2500000.0           # MI355X_MICROARCH.md: dense f16 / bf16 MFMA peak (no sparsity)
HBM_PEAK_GBS = 8000.0


def fixture_weights(model_name, net, kind):
    """'he': He-scaled seeded weights (denoisers.seeded_state_dict, seed 1: the weights of every PnP line of rounds 2-4);
    'contractive': the fixture weights of the 50-iteration goldens (tests/golden/pnp_known.json: seeds + operator-norm gains), with
    which D is a contraction, the loop is stable, and the `parity` record below measures the implementation instead of the chaos
    of an expansive random network.  Same architecture, same arithmetic, same launches either way."""
    from pnp_admm_cnc_mri_amd import denoisers as D
    if kind == 'he':
        return D.seeded_state_dict(net, 1), None
    if kind == 'trained':                               # FFDNet-gray only: the network trained by oracle/train_fixture_denoiser.py
        import torch
        if model_name != 'ffdnet_gray':
            raise SystemExit("--weights trained: only ffdnet_gray has a trained fixture (tests/golden/ffdnet_gray_trained.npz)")
        w = np.load(os.path.join(ROOT, 'tests', 'golden', 'ffdnet_gray_trained.npz'))
        return {k: torch.from_numpy(w[k]) for k in w.files}, None
    meta = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'pnp_known.json')))
    seed, gains, fam = meta['known50']['seeds'][model_name], meta['gains50'], D.family(model_name)
    if fam == 'ircnn':
        bank = {str(i): D.contractive_state_dict(net, fam, seed + i, gains['%s/%d' % (model_name, i)]) for i in range(25)}
        return bank['0'], bank
    return D.contractive_state_dict(net, fam, seed, gains[model_name]), None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='ffdnet_gray')
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--batch', type=int, default=512, help='slices per GPU')
    ap.add_argument('--size', type=int, default=256, choices=[256, 512], help='512 = the shape of config 5 (seeded mask bank of 3)')
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--cnn-batch', type=int, default=None, help='slices per CNN call (default: the Denoiser\'s: 64, 256 for the plain stacks on the HIP backends)')
    ap.add_argument('--mask', default=None)
    ap.add_argument('--cnn-dtype', default=None, choices=[None, 'bf16', 'fp16'], help='autocast throughput mode, off parity')
    ap.add_argument('--cnn-backend', default='torch', choices=['torch', 'hip', 'hip_f16x3', 'auto'],
                    help="hip = the plain stacks' conv3x3 layers (FFDNet, DnCNN, IRCNN) and DRUNet's 64-channel blocks on libpnpmri.so's "
                         "fp32-MFMA kernel; hip_f16x3 = the same with the 64 -> 64 layers in split-half arithmetic on the f16 matrix cores")
    ap.add_argument('--cnn-graph', action='store_true', help='replay each denoiser forward (<= cnn-batch slices) from a captured HIP graph')
    ap.add_argument('--miopen-find', action='store_true', help='torch.backends.cudnn.benchmark = True for the whole run')
    ap.add_argument('--launch-timeout', type=float, default=float(os.environ.get('PNP_BENCH_TIMEOUT', 1500)))
    ap.add_argument('--rehearse-gloo', action='store_true', help='N > 1 on a box with ONE GPU: gloo backend, all ranks on cuda:0')
    ap.add_argument('--weights', default='auto', choices=['auto', 'contractive', 'he', 'trained'],
                    help="'auto' (default): 'trained' where a trained fixture exists (ffdnet_gray), 'contractive' otherwise.  'contractive' (the 50-iteration goldens' fixture weights: a stable loop, a meaningful "
                         "parity record), 'he' (He-scaled random weights: the lines of rounds 2-4) or 'trained' (ffdnet_gray: the network trained by "
                         "oracle/train_fixture_denoiser.py -- a working denoiser's activation statistics)")
    ap.add_argument('--sustain-s', type=float, default=0.0, help='seconds of the `sustained` sub-record (N = 1; 0 = none)')
    ap.add_argument('--no-parity', action='store_true', help='skip the oracle loop on three slices (N = 1 only; a few CPU seconds)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        import bench                                   # stdlib + numpy only at import: the parent never touches the GPU
        sys.exit(bench.launch_ranks(args.gpus, args.launch_timeout, script=__file__))

    rank = int(os.environ.get('RANK', '0'))
    local_rank = 0 if args.rehearse_gloo else int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    import bench
    out = bench.claim_stdout()                       # ONE JSON line on stdout, whatever RCCL / MIOpen print

    import torch
    import pnp_admm_cnc_mri_amd as P
    from pnp_admm_cnc_mri_amd import denoisers as D, solvers_pnp as SP, synthetic as S, utils_pnp, sharding

    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get('PNP_BENCH_FORCE_DIST') == '1':       # the env hook exercises RCCL with one rank (as in bench.py)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.rehearse_gloo:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))

    fam = D.family(args.model)
    if args.weights == 'auto':          # a working denoiser's activation statistics where the repo has one (they set the clock the f16x3 kernel holds)
        args.weights = 'trained' if args.model == 'ffdnet_gray' else 'contractive'
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    H = W = args.size
    B = args.batch
    if H == 256:
        mname = args.mask or {'ffdnet': 'Q_Radial30', 'drunet': 'Q_Cartesian30'}.get(fam, 'Q_Random30')
        masks = S.reference_masks()[mname].astype(np.uint8)[None]
        mask_id = np.zeros(B, np.int32)
    else:
        mname = 'seeded bank (random, radial, cartesian)'
        masks = np.stack([S.synthetic_mask(k, H, W) for k in ('random', 'radial', 'cartesian')])
        mask_id = (np.arange(B) % 3).astype(np.int32)
    img, noise = S.batch(rank * B, B, H, W)             # this rank's shard of the job
    opts = SP.PRESETS['PNP_ADMM_CNC_D'].get(fam, SP.PRESETS['PNP_ADMM_CNC_DnCNN'])
    dev = torch.device('cuda', local_rank)
    net, nlm, sched = D.build(args.model)
    sd, bank = fixture_weights(args.model, net, args.weights)
    net.load_state_dict(sd)
    iters = args.warmup + args.steps
    sig = None
    if sched:
        sig = torch.tensor(utils_pnp.get_rho_sigma(max(0.255 / 255., nlm), iters, 49, nlm * 255., 1.0)[1])
    den = D.Denoiser(args.model, net.eval(), nlm, sigmas=sig, noises=noise[0], bank=bank, cnn_batch=args.cnn_batch, cnn_dtype=args.cnn_dtype,
                     backend=args.cnn_backend, graph=args.cnn_graph).to(dev)
    args.cnn_backend = den.backend                      # 'auto' resolved (denoisers.auto_backend)
    args.cnn_batch = den.cnn_batch                      # None resolved
    flop_per_call, flop_f16x3_part = D.forward_flops(den, H, W, dev, detail=True)     # one slice, one D(.)

    eng = P.Engine(H, W, Bmax=B, device=local_rank)
    eng.synthesize(img, noise, masks, mask_id)
    eng.init_state()
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    z = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
    w = torch.empty_like(z)
    eng.get_state(z, w)
    x, s, t, zn = (torch.empty_like(z) for _ in range(4))
    den_finite = torch.ones((), dtype=torch.bool, device=dev)      # every denoiser output of the run, BEFORE the clamp, stayed finite
    # HIP events per iteration, READ AFTER THE LOOP: nothing on the host waits for the device between the opening and the closing sync
    # (round 5 synchronised inside every iteration to read its events -- a host round trip per step of a kernel train whose clock is set by
    # sustained power is not a sustained rate)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(iters)]

    def one_iteration(i, ev=None):
        nonlocal z, zn, den_finite
        if ev:
            ev[0].record()
        eng.dc_step(z, w, x, opts['reo'])
        if ev:
            ev[1].record()
        den.select_bank(i)
        den(z, i, out=s)
        eng.cnc_combine(z, x, w, s, t, opts['alpha'], opts['lambda1'], opts['reo'], opts['b'])
        den(t, i, out=zn)
        if ev:
            ev[2].record()
        den_finite &= torch.isfinite(s).all() & torch.isfinite(zn).all()
        eng.dual_clamp(x, zn, w)
        if ev:
            ev[3].record()
        z, zn = zn, z
    t0 = time.perf_counter()
    with torch.no_grad():
        for i in range(iters):
            if i == args.warmup:
                torch.cuda.synchronize()
                if dist is not None:
                    dist.barrier()
                    torch.cuda.synchronize()
                t0 = time.perf_counter()
            one_iteration(i, evs[i])
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0              # this rank's clock stops at its own sync, before any collective
    t_dc = sum(e[0].elapsed_time(e[1]) + e[2].elapsed_time(e[3]) for e in evs[args.warmup:])
    t_cnn = sum(e[1].elapsed_time(e[2]) for e in evs[args.warmup:])
    x_timed = x.clone()                               # what the parity leg checks: x after exactly `iters` iterations

    # Checker leg (N = 1): the oracle's PnP loop (float64 NumPy x-update, the reference's marshalling) on three slices of the run that
    # was just timed, driven by the SAME denoiser object one slice per call -- what differs from the timed run is the HIP x-update /
    # glue kernels and the batch size of the CNN calls.
    parity = None
    if world == 1 and not args.no_parity:
        from oracle import admm_oracle as O
        picks = sorted({0, max(B // 2 - 1, 0), B - 1})
        y_all = eng.download_y()

        def denoise(a, i):
            den.select_bank(i)
            tt = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))[None, None].to(dev)
            return den(tt, i)[0, 0].cpu().numpy()
        rels = []
        with torch.no_grad():
            for b in picks:
                ref = O.pnp_admm_cnc(y_all[b].astype(np.complex128), masks[mask_id[b]], denoise, iters, opts['alpha'], opts['lambda1'], opts['reo'], opts['b'])
                got = x_timed[b, 0].double().cpu().numpy()
                rels.append(float(np.linalg.norm(got - ref) / np.linalg.norm(ref)))
        parity = {'rel_l2_vs_oracle': rels, 'slices': picks, 'iterations': iters,
                  'oracle': 'oracle/admm_oracle.py pnp_admm_cnc (NumPy float64 x-update) driven by the same denoiser, one slice per call'}
        del y_all

    # `sustained` (N = 1): the same iteration back to back for >= --sustain-s seconds, one host sync per chunk of `iters` iterations, HIP
    # events over the span -- the rate the conv kernels hold at the clock the board settles on (the K timed steps above are ~0.1 .. 2 s
    # from a cold process).  The state simply keeps iterating (sigma / bank index of the last step): values stay finite behind the clamp.
    sustained = None
    if world == 1 and args.sustain_s > 0:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 0
        with torch.no_grad():
            e0.record()
            while True:
                for _ in range(max(iters, 2)):
                    one_iteration(iters - 1)
                n += max(iters, 2)
                e1.record()
                e1.synchronize()
                if e0.elapsed_time(e1) >= args.sustain_s * 1e3:
                    break
        span = e0.elapsed_time(e1) / 1e3
        sustained = {'value': n / span, 'steps': n, 'span_s': span, 'ms_per_step': span * 1e3 / n,
                     'note': 'back-to-back iterations, one host sync per %d, HIP events over the span' % max(iters, 2)}

    gather_ms = None
    if dist is not None:
        dist.barrier()
        xg = x.reshape(B, H, W)
        xg = xg.cpu() if args.rehearse_gloo else xg
        sharding.gather_slices(xg, world * B, dst=0)                   # untimed: connection set-up
        torch.cuda.synchronize()
        dist.barrier()
        tg = time.perf_counter()
        x_all = sharding.gather_slices(xg, world * B, dst=0)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg) * 1e3
        assert (x_all is not None) == (rank == 0)
        mine = torch.tensor([wall, t_dc, t_cnn, gather_ms], dtype=torch.float64, device='cpu' if args.rehearse_gloo else 'cuda')
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)                                    # every rank's own clocks: a slow GPU shows in the line
        per_rank = torch.stack(every).cpu()
        wall, t_dc, t_cnn, gather_ms = (float(v) for v in per_rank.max(dim=0).values)   # the job's time: MAX over ranks
    if rank == 0:
        K = args.steps
        dc_ms, cnn_ms = t_dc / K, t_cnn / K
        den_flop = 2.0 * flop_per_call * B                              # two D(.) per iteration, every slice
        den_gflops = den_flop / (cnn_ms * 1e-3) / 1e9
        # What the matrix pipe really ISSUES: under 'hip_f16x3' a float32 product of the C -> C layers (and of the last layer) is three
        # f16 matrix products, and the roof over those is the dense f16 peak; the remaining layers (the first layer on the vector
        # units, DRUNet's strided / transposed convolutions) count once.  The float32-equivalent figure stays as frac_fp32_equivalent.
        f16x3 = args.cnn_backend == 'hip_f16x3'
        issued_flop = 2.0 * B * (3.0 * flop_f16x3_part + (flop_per_call - flop_f16x3_part)) if f16x3 else den_flop
        issued_gflops = issued_flop / (cnn_ms * 1e-3) / 1e9
        den_peak = F16_MATRIX_PEAK_GFLOPS if f16x3 else F32_MATRIX_PEAK_GFLOPS
        alg_bytes = 57.0 * H * W * B
        out.write(json.dumps({
            'metric': 'PNP_ADMM_CNC_D iterations/sec on %dx%d slices (%s)' % (H, W, args.model),
            'value': world * K / wall * (B / 512.0), 'unit': 'it/s (512-slice batches)', 'n_gpus': world, 'steps': K,
            'warmup': args.warmup, 'ms_per_step': wall / K * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'dtype': ('f32' if args.cnn_backend != 'hip_f16x3' else 'f32 (conv 64->64: f32 operands as half pairs, exact products, f32 accumulation)') if args.cnn_dtype is None else args.cnn_dtype,
            'data': 'synthetic (seeded %s weights)' % args.weights,
            'config': {'workload': 'PNP_ADMM_CNC_D, %s, %d synthetic %dx%d slices per GPU, %s, S6:569-577 presets'
                                   % (args.model, B, H, W, mname), 'slices_per_gpu': B, 'path': eng.path_name,
                       'cnn_batch': args.cnn_batch, 'cnn_backend': args.cnn_backend, 'cnn_graph': bool(args.cnn_graph), 'weights': args.weights},
            'parity': parity, 'sustained': sustained,
            'slice_iterations_per_s': world * K * B / wall, 'gather_ms': gather_ms,
            'per_rank': None if dist is None else {'ms_per_step': [float(v) / K * 1e3 for v in per_rank[:, 0]],
                                                   'gather_ms': [float(v) for v in per_rank[:, 3]]},
            'fft_prox': {'ms_per_step': dc_ms, 'share': dc_ms / (dc_ms + cnn_ms), 'algorithmic_bytes': alg_bytes,
                         'roofline': {'bound': 'hbm', 'achieved': alg_bytes / (dc_ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                      'frac': alg_bytes / (dc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                         'note': 'pnp_dc_step + pnp_cnc_combine is inside the denoiser span; pnp_dual_clamp is here; '
                                 '57 N bytes per slice-iteration is the contract figure of the whole FFT + prox iteration'},
            'denoiser': {'ms_per_step': cnn_ms, 'share': cnn_ms / (dc_ms + cnn_ms), 'calls_per_step': 2,
                         'flop_per_call_per_slice': flop_per_call, 'flop_per_step': den_flop,
                         'roofline': {'bound': 'mfma_f16' if f16x3 else 'mfma_f32', 'achieved': issued_gflops, 'peak': den_peak, 'unit': 'GFLOP/s',
                                      'frac': issued_gflops / den_peak, 'issued_flop_per_step': issued_flop,
                                      'achieved_fp32_equivalent': den_gflops, 'frac_fp32_equivalent': den_gflops / F32_MATRIX_PEAK_GFLOPS},
                         'note': 'PyTorch-ROCm / MIOpen fp32 convolutions (north star: PyTorch for the CNN forward)' if args.cnn_backend == 'torch' else
                                 'body layers (64 -> 64 conv3x3 + ReLU) on the fp32-MFMA implicit GEMM of libpnpmri.so (kernels_conv.hip); first / last layer on its direct kernels' if args.cnn_backend == 'hip' else
                                 'C -> C conv3x3 layers in split-half arithmetic on the f16 matrix cores (kernels_conv_f16x3.hip): roofline = the f16 matrix products really issued (3 per float32 product) against the dense f16 peak; frac_fp32_equivalent prices the float32-equivalent arithmetic against the FLOAT32 matrix peak and can exceed 1'},
            'x_finite': bool(torch.isfinite(x_timed).all()),
            'denoiser_outputs_finite': bool(den_finite)}) + '\n')      # taken BEFORE pnp_dual_clamp, every iteration (warm-up included)
        out.flush()
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
