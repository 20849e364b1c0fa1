#!/usr/bin/env python3
"""Headline benchmark: ADMM iterations/s on 256x256 complex64 slices, batch = 512 per GPU
(BASELINE.json configs[1]: ADMM_CNC, Q_Random30, S4:176 presets), synthetic inputs of
SURVEY.md section 8(d).

    python bench.py --gpus N --steps K --warmup W

One "step" = one batched ADMM iteration (x-update in k-space, CNC z-update, dual update) over
the 512 slices a GPU holds.  Slices are independent, so N GPUs hold N x 512 different slices
(weak scaling), run without any data-path collective, and one RCCL gather of x at the end is
timed separately (`gather_ms`).  Inputs are resident in HBM before the timed region.
`python bench.py --gpus N` starts its own N rank processes (children of a parent that never
touches the GPU); under `torch.distributed.run` it takes the ranks it is given.

At 256x256 with a chip-filling batch the K steps of a call are ONE launch of the slice-resident
kernel (config.path = "slice": a workgroup keeps its slice in registers for all K iterations);
smaller batches, 512x512 and --precision f64 use the two-launch fused kernels ("fused").

Prints ONE JSON line (rank 0).  `value` = 512-slice batch iterations per second summed over
ranks = slice-iterations/s / 512.  `roofline.achieved` / `roofline.frac` (schema 2, since round 4) =
the HBM bytes the kernels REALLY move per batched iteration (rocprofv3 PMC of exactly this
configuration, committed in profiles/traffic.json; the kernel family's own algorithmic bytes
where no PMC figure exists) / HIP-event time per iteration on the kernels' own stream -- a
physical fraction of the 8 TB/s peak, <= 1.  `roofline.frac_contract_57N` prices the same time
against SURVEY.md 8(d)'s 57*N_pix*B bytes (the quantity rounds 1-3 printed under `frac`; it
exceeds 1 because the kernels move fewer bytes than that formulation).
Sub-records at N = 1: `sustained` (back-to-back 100-iteration solves), `l1` (configs[0]'s solver,
ADMM_L1, on the same batch), `f64` (the double-precision engine), `pnp` (configs[2] through
bench_pnp.py, two child processes after this one has released the GPU), `parity` (three slices of
the timed run against the NumPy oracle), `cpu_baseline` = the NumPy oracle (float64, np.fft,
1 thread) on a bounded sample of the same slices.
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

H = W = 256
B_PER_GPU = 512
PRESET = dict(alpha=0.45, lambda1=0.5, reo=0.05, b=64)          # S4:176
ALG_BYTES_PER_PIXEL = 57
HBM_PEAK_GBS = 8000.0                                              # MI355X_MICROARCH.md
# Bytes per pixel and slice-iteration each kernel family must move BY DESIGN (DESIGN.md sections 3, 4), float32;
# a double-precision context doubles them.  slice: z, w read + written (16) + Hermitian half-plane table (4), ADMM_L1
# single-state form 8 + 4; fused 256: + the transposed field of two slices per complex transform out and back (16);
# fused 512: 36.4 measured layout; generic: the plain c2c contract.
OWN_BYTES_PER_PIXEL = {('slice', 'cnc'): 20.0, ('slice', 'l1'): 12.0, ('fused', 'cnc'): 36.0, ('fused', 'l1'): 28.0,
                       ('generic', 'cnc'): 57.0, ('generic', 'l1'): 57.0}


def cpu_baseline(masks, mask_id, budget_s=20.0, iters=100):
    """Oracle timing on this host: as many whole 100-iteration slice solves as fit the budget."""
    from oracle import admm_oracle as O
    from pnp_admm_cnc_mri_amd import synthetic as S
    t_used, n_slices = 0.0, 0
    # one untimed warm-up slice-solve of 5 iterations (imports, pocketfft plan cache)
    n_avail = len(mask_id)
    y = O.synthesize(S.phantom(0, H, W), masks[mask_id[0]].astype(np.float64), S.kspace_noise(0, H, W))
    O.admm_cnc(y, masks[mask_id[0]], 5, **PRESET)
    while t_used < budget_s and n_slices < n_avail:
        b = n_slices
        y = O.synthesize(S.phantom(b, H, W), masks[mask_id[b]].astype(np.float64), S.kspace_noise(b, H, W))
        t0 = time.perf_counter()
        O.admm_cnc(y, masks[mask_id[b]], iters, **PRESET)
        t_used += time.perf_counter() - t0
        n_slices += 1
    slice_it_s = n_slices * iters / t_used
    return {'value': slice_it_s / B_PER_GPU, 'unit': 'it/s (512-slice batches)', 'cores': 1, 'kind': 'port',
            'slice_iterations_per_s': slice_it_s,
            'sample': '%d slices x %d iterations of the same workload, %.1f s, NumPy %s float64 np.fft, 1 thread'
                      % (n_slices, iters, t_used, np.__version__)}


def _cpu_worker(args):
    """one worker of the all-cores baseline: whole 100-iteration solves of its own slices for `budget` s"""
    first, stride, budget, iters, size = args
    import numpy as _np
    from oracle import admm_oracle as O
    from pnp_admm_cnc_mri_amd import synthetic as S
    mask = (S.reference_masks()['Q_Random30'] if size == 256 else S.synthetic_mask('random', size, size)).astype(_np.uint8)
    t_used, n, b = 0.0, 0, first
    while t_used < budget:
        y = O.synthesize(S.phantom(b, size, size), mask.astype(_np.float64), S.kspace_noise(b, size, size))
        t0 = time.perf_counter()
        O.admm_cnc(y, mask, iters, **PRESET)
        t_used += time.perf_counter() - t0
        n += 1
        b += stride
    return n, t_used


def cpu_baseline_all_cores(budget_s=8.0, iters=100):
    """one worker process per host core, each looping over its own slices."""
    import multiprocessing as mp
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))                        # a one-GPU box's CPU share is 16 cores
    os.environ.setdefault('OMP_NUM_THREADS', '1')
    with mp.get_context('spawn').Pool(cores) as pool:
        t0 = time.perf_counter()
        res = pool.map(_cpu_worker, [(w, cores, budget_s, iters, H) for w in range(cores)])
        wall = time.perf_counter() - t0
    slice_it_s = sum(n * iters / t for n, t in res)          # sum of per-worker rates (start-up excluded)
    return {'value': slice_it_s / B_PER_GPU, 'unit': 'it/s (512-slice batches)', 'cores': cores, 'kind': 'port',
            'slice_iterations_per_s': slice_it_s,
            'sample': '%d worker processes x %.0f s of whole %d-iteration slice solves (%d slices), wall %.1f s'
                      % (cores, budget_s, iters, sum(n for n, _ in res), wall)}


def launch_ranks(n, timeout_s, script=None):
    """`python bench.py --gpus N` outside torchrun: start the N rank processes as CHILDREN of this
    process (which has not imported torch and never initialises HIP), one per GPU, with the
    torch.distributed env of a one-node job on 127.0.0.1; relay rank 0's JSON line and return the
    worst exit code.  No exec of an initialised process, no hop under a profiler.
    A rank that dies takes the job down; a rank that hangs (rendezvous, collective) does so until
    `timeout_s`, then every child is terminated, killed if need be, and the exit code is 124."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    procs, outs = [], []
    deadline = time.monotonic() + timeout_s
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        # RCCL shares device buffers between the rank processes of a node through HIP IPC handles.  The host
        # driver of this pool supports only the dmabuf form (hipIpcGetMemHandle fails with "invalid argument"
        # in legacy mode), which HSA_ENABLE_IPC_MODE_LEGACY=0 selects; the image exports it, a caller's own
        # value wins.
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        outs.append(tempfile.TemporaryFile())
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script or __file__)] + sys.argv[1:], env=env, stdout=outs[r]))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        if time.monotonic() > deadline:
            sys.stderr.write('bench.py: ranks still running after %.0f s: terminating them\n' % timeout_s)
            for q in live:
                q.terminate()
            t_kill = time.monotonic() + 10
            while any(q.poll() is None for q in live) and time.monotonic() < t_kill:
                time.sleep(0.1)
            for q in live:
                if q.poll() is None:
                    q.kill()
            rc = 124
            break
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = rc or code
                for q in live:
                    q.terminate()
    for p in procs:
        try:
            p.wait(timeout=15)
        except Exception:
            p.kill()
    for r, fo in enumerate(outs):
        fo.seek(0)
        text = fo.read().decode(errors='replace')
        fo.close()
        if r == 0:
            sys.stdout.write(text)
            sys.stdout.flush()
        elif rc != 0 and text:                          # what a failed job's other ranks printed
            sys.stderr.write('--- stdout of rank %d ---\n%s\n' % (r, text))
    return rc


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner when its
    process group comes up), so everything that is not the line goes to stderr: file descriptor 1 is pointed at stderr for the
    rest of the process and the returned file object is the only way to the real stdout."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    return real


def latency_record(device=0):
    """The reference's OWN usage as a driver-visible number: ONE 256 x 256 slice, the committed 50 iterations (S4:83-138; S6:231-308) -- ms per
    solve with the inputs on the device, HIP events around the loop, best of three after one warm-up solve, each with its relative L2 distance
    from the float64 oracle on the same measurements.  BASELINE.md section 1 quotes ~0.46 s / ~0.48 s per image for the reference's ADMM_L1 /
    ADMM_CNC and ~4.1 s for its PnP pair (log timestamps of the authors' machine: context, not a same-node comparison).
    What bounds a one-slice solve: ONE compute unit of 256 is busy (the slice-resident kernel keeps a slice in one unit's registers; the
    PnP forwards are 64 .. 256 tiles on 256 units, a train of 17 .. 70 launches per forward) -- latency, not bandwidth."""
    import torch
    import pnp_admm_cnc_mri_amd as P
    from pnp_admm_cnc_mri_amd import denoisers as D, solvers_pnp as SP, synthetic as S, utils_pnp
    from oracle import admm_oracle as O
    import bench_pnp
    rec = {'unit': 'ms per 50-iteration solve of one 256x256 slice', 'iterations': 50,
           'reference_context': 'BASELINE.md: ~460 / ~480 ms per image (ADMM_L1 / ADMM_CNC), ~4100 ms (PNP_ADMM_CNC_DnCNN), authors\' log timestamps'}
    dev = torch.device('cuda', device)
    mk = S.reference_masks()
    img, noise = S.batch(0, 1)

    def best_of(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        best = None
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1)
            best = ms if best is None else min(best, ms)
        return best
    # plain loops through the engine (ADMM_L1: S1:171 presets; ADMM_CNC: S4:176 presets)
    mask = mk['Q_Random30'].astype(np.uint8)
    for key, prec in (('admm_l1_f32', 'f32'), ('admm_cnc_f32', 'f32'), ('admm_cnc_f64', 'f64')):
        with P.Engine(256, 256, Bmax=1, device=device, precision=prec) as eng:
            eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
            eng.synthesize(img, noise, mask[None])
            y = eng.download_y()

            def solve():
                eng.init_state()
                if 'l1' in key:
                    eng.admm_l1(50, 0.1, 0.015)
                else:
                    eng.admm_cnc(50, PRESET['alpha'], PRESET['lambda1'], PRESET['reo'], PRESET['b'])
            ms = best_of(solve)
            x = eng.x()[0].astype(np.float64)
            ref = (O.admm_l1(y[0].astype(np.complex128), mask, 50, lambda1=0.1, reo=0.015) if 'l1' in key else
                   O.admm_cnc(y[0].astype(np.complex128), mask, 50, PRESET['alpha'], PRESET['lambda1'], PRESET['reo'], PRESET['b']))
            rec[key] = {'ms': ms, 'rel_l2_vs_oracle': float(np.linalg.norm(x - ref) / np.linalg.norm(ref)), 'path': eng.path_name}
    # PnP: PNP_ADMM_CNC_D's loop body (solvers_pnp.py:126-133) on one slice, split-half backend, with and without HIP-graph replay of the forwards
    for model, mname in (('ffdnet_gray', 'Q_Radial30'), ('drunet_gray', 'Q_Cartesian30')):
        fam = D.family(model)
        opts = SP.PRESETS['PNP_ADMM_CNC_D'][fam]
        m = mk[mname].astype(np.uint8)
        net, nlm, sched = D.build(model)
        sd, _ = bench_pnp.fixture_weights(model, net, 'contractive')
        net.load_state_dict(sd)
        sig = torch.tensor(utils_pnp.get_rho_sigma(max(0.255 / 255., nlm), 50, 49, nlm * 255., 1.0)[1]) if sched else None
        for graph in (False, True):
            den = D.Denoiser(model, net.eval(), nlm, sigmas=sig, backend='hip_f16x3', graph=graph).to(dev)
            with torch.no_grad(), P.Engine(256, 256, Bmax=1, device=device) as eng:
                eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
                eng.synthesize(img, noise, m[None])
                z = torch.empty((1, 1, 256, 256), device=dev)
                w = torch.empty_like(z)
                x, s_, t_, zn = (torch.empty_like(z) for _ in range(4))
                st = {}

                def solve():
                    eng.init_state()
                    eng.get_state(z, w)
                    a, b = z, zn
                    for i in range(50):
                        eng.dc_step(a, w, x, opts['reo'])
                        den(a, i, out=s_)
                        eng.cnc_combine(a, x, w, s_, t_, opts['alpha'], opts['lambda1'], opts['reo'], opts['b'])
                        den(t_, i, out=b)
                        eng.dual_clamp(x, b, w)
                        a, b = b, a
                    st['x'] = x
                ms = best_of(solve)
                got = st['x'][0, 0].double().cpu().numpy()
                y = eng.download_y()
                r = {'ms': ms}
                if not graph:                                          # (the graph replays the same kernels: one oracle loop per model)
                    def denoise(a_, i):
                        return den(torch.from_numpy(np.ascontiguousarray(a_, dtype=np.float32))[None, None].to(dev), i)[0, 0].cpu().numpy()
                    ref = O.pnp_admm_cnc(y[0].astype(np.complex128), m, denoise, 50, opts['alpha'], opts['lambda1'], opts['reo'], opts['b'])
                    r['rel_l2_vs_oracle'] = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
                rec['pnp_cnc_d_%s_hip_f16x3%s' % (model, '_graph' if graph else '')] = r
        del net
    return rec


def pnp_record(steps=3, warmup=1, total_timeout=300):
    """BASELINE.json configs[2] (PNP_ADMM_CNC_D, FFDNet-gray, 512 slices of 256 x 256, Q_Radial30) measured by bench_pnp.py in two
    child processes -- plus configs[3]'s per-GPU shard (DRUNet-gray, Q_Cartesian30) on the f16x3 backend in a third -- AFTER this process
    has released its engine and buffers: the CNN forward on PyTorch-ROCm / MIOpen (the north
    star's split) and on libpnpmri.so's split-half f16 matrix-core kernels (`cnn_backend='hip_f16x3'`).  Each child's
    line carries its own parity record (three slices of the timed run against the oracle's loop) and its physical roofline (the matrix
    products really issued against the peak of the pipe that runs them).  A sub-record: it never fails the main line; the children
    share ONE time budget, and the tail of a failed child's stderr is kept."""
    import subprocess
    rec = {'config': 'configs[2]: PNP_ADMM_CNC_D, FFDNet-gray, 512 x 256x256 slices, Q_Radial30, S6:573 presets (bench_pnp.py --steps %d --warmup %d)'
                     % (steps, warmup), 'unit': 'it/s (512-slice batches)'}
    deadline = time.monotonic() + total_timeout
    # third child: BASELINE.json configs[3]'s per-GPU shard (DRUNet-gray, 512 slices, Q_Cartesian30) on the f16x3 backend -- every layer of the
    # U-Net on libpnpmri.so; its PyTorch / MIOpen counterpart needs a two-minute find on a fresh box and stays with bench_pnp.py
    for key, model, backend, st in (('torch', 'ffdnet_gray', 'torch', steps), ('hip_f16x3', 'ffdnet_gray', 'hip_f16x3', steps),
                                    ('config4_shard_drunet_hip_f16x3', 'drunet_gray', 'hip_f16x3', max(1, steps - 1))):
        err = b''
        try:
            left = deadline - time.monotonic()
            if left < 20:
                raise TimeoutError('the record\'s time budget of %d s is spent' % total_timeout)
            r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench_pnp.py'), '--model', model, '--batch', '512', '--steps', str(st),
                                '--warmup', str(warmup), '--cnn-backend', backend, '--sustain-s', '0' if backend == 'torch' else os.environ.get('PNP_BENCH_PNP_SUSTAIN_S', '5')],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               timeout=left, cwd=ROOT)
            err = r.stderr
            j = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith('{')][-1])
            rf = j['denoiser']['roofline']
            rec[key] = {'value': j['value'], 'ms_per_step': j['ms_per_step'], 'denoiser_ms_per_step': j['denoiser']['ms_per_step'],
                        'denoiser_roofline': {k: rf[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'frac_fp32_equivalent')},
                        'parity': j['parity'], 'sustained': j.get('sustained'), 'weights': j['config']['weights'], 'workload': j['config']['workload'],
                        'x_finite': j['x_finite'], 'denoiser_outputs_finite': j['denoiser_outputs_finite']}
        except subprocess.TimeoutExpired as e:
            rec[key] = {'error': 'timeout', 'stderr_tail': (e.stderr or b'').decode(errors='replace')[-600:]}
        except Exception as e:                                       # noqa: BLE001 -- a failed child is reported, not raised
            rec[key] = {'error': repr(e)[:300], 'stderr_tail': err.decode(errors='replace')[-600:]}
    if 'value' in rec.get('torch', {}) and 'value' in rec.get('hip_f16x3', {}):
        rec['speedup'] = rec['hip_f16x3']['value'] / rec['torch']['value']
    rec['note'] = ('hip_f16x3: float32 operands carried as two halves, three exact-product f16 matrix instructions per product, float32 '
                   'accumulation -- per-layer error below the PyTorch / MIOpen float32 kernels\' (tests/test_gpu_conv.py).  denoiser_roofline.frac: '
                   'matrix products really issued (3 per float32 product under hip_f16x3) / time / the peak of the pipe that runs them (dense f16: '
                   '2.5 PFLOP/s; fp32: 157.3 TFLOP/s); frac_fp32_equivalent prices the float32-equivalent arithmetic against the float32 peak '
                   'and can exceed 1')
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=B_PER_GPU, help='slices per GPU (default 512 = the headline config)')
    ap.add_argument('--solver', choices=['cnc', 'l1'], default='cnc')
    ap.add_argument('--generic', action='store_true', help='force the generic (unfused) kernels')
    ap.add_argument('--size', type=int, default=256, choices=[256, 512],
                    help='slice edge; 512 = the shape of config 5 (seeded masks, mask_id = b %% 3), not the headline')
    ap.add_argument('--precision', choices=['f32', 'f64'], default='f32',
                    help="f64 = the double-precision engine (meets 1e-5 against the float64 reference at 100 CNC "
                         "iterations, DESIGN.md section 2); the headline stays f32")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget', type=float, default=20.0)
    ap.add_argument('--launch-timeout', type=float, default=float(os.environ.get('PNP_BENCH_TIMEOUT', 1500)),
                    help='--gpus N without torchrun: seconds after which hanging rank processes are killed (exit 124)')
    ap.add_argument('--no-pnp-record', action='store_true',
                    help='skip the `pnp` sub-record (configs[2]: FFDNet PnP on the PyTorch / MIOpen backend and on the f16x3 HIP backend, configs[3] shard: DRUNet on the f16x3 backend; three child runs of bench_pnp.py, ~60 s)')
    ap.add_argument('--no-latency-record', action='store_true',
                    help='skip the `latency` sub-record (one 256 x 256 slice, 50 iterations: ADMM_L1, ADMM_CNC f32 / f64, PnP with FFDNet and DRUNet; ~8 s)')
    ap.add_argument('--no-l1-record', action='store_true',
                    help="skip the `l1` sub-record (ADMM_L1, configs[0]'s solver with the S1:171 presets, on the same batch; N = 1, headline configuration only)")
    ap.add_argument('--no-f64-record', action='store_true',
                    help="skip the double-precision engine's sub-record (N = 1, headline configuration only)")
    ap.add_argument('--sustain-s', type=float, default=2.0,
                    help='seconds of the `sustained` sub-record: back-to-back calls of --steps iterations for this long '
                         '(after half as long a pre-heat), HIP events over the whole span; 0 = off')
    ap.add_argument('--sustain-steps', type=int, default=100,
                    help='sustained record: iterations per call (default 100 = the run length of BASELINE.json configs[1])')
    ap.add_argument('--sustain-reinit', type=int, default=1,
                    help='sustained record: 1 (default) = every call starts from a fresh state z0 = |ifft2(y)|, w0 = 0, as a job '
                         'that solves batch after batch does (the initialisation kernels are inside the span, only iterations '
                         'are counted); 0 = keep iterating the same state (it converges: data with many exact zeros)')
    ap.add_argument('--rehearse-gloo', action='store_true',
                    help='rehearsal of the N>1 launch path on a box with ONE GPU: gloo backend, all ranks on cuda:0')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus, args.launch_timeout))        # parent: never touches the GPU

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        args.gpus = world
    out = claim_stdout()

    import torch
    import pnp_admm_cnc_mri_amd as P
    from pnp_admm_cnc_mri_amd import synthetic as S

    if args.rehearse_gloo:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get('PNP_BENCH_FORCE_DIST') == '1':     # the env hook exercises RCCL with one rank
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.rehearse_gloo:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))

    B = args.batch
    global H, W
    H = W = args.size
    if args.size == 256:
        mk = S.reference_masks()
        masks = np.stack([mk['Q_Random30']]).astype(np.uint8)        # config 2: Q_Random30 for every slice
        mask_id = np.zeros(B, np.int32)
    else:
        masks = np.stack([S.synthetic_mask(k, H, W) for k in ('random', 'radial', 'cartesian')])
        mask_id = (np.arange(B) % 3).astype(np.int32)
    cache = os.environ.get('PNP_BENCH_CACHE')                         # repeated runs: reuse the generated inputs
    cpath = cache and '%s.r%d.b%d.n%d.npz' % (cache, rank, B, H)
    if cpath and os.path.exists(cpath):
        d = np.load(cpath)
        img, noise = d['img'], d['noise']
    else:
        img, noise = S.batch(rank * B, B, H, W)                       # this rank's shard of the job
        if cpath:
            np.savez(cpath, img=img, noise=noise)

    eng = P.Engine(H, W, Bmax=B, device=local_rank)
    if args.generic:
        eng.set_fast_path(0)
    eng.synthesize(img, noise, masks, mask_id)                        # y = fft2(img)*mask + noise, on device
    if args.precision == 'f64':                                       # same measurements, double-precision engine
        y32 = eng.download_y()
        eng.close()
        eng = P.Engine(H, W, Bmax=B, device=local_rank, precision='f64')
        if args.generic:
            eng.set_fast_path(0)
        eng.upload(y32, masks, mask_id)
        del y32
    eng.init_state()
    eng.prepare_loops()                                               # per-problem tables of the loop kernels: never inside a timed region
    # Attribution (round 6): which card, at which nominal clock, and what ITS memory system gives a plain streaming kernel with the slice
    # kernel's access shape today (pnp_calibrate_stream, >= 0.6 s, before anything is timed) -- boxes of the pool differ by several per cent
    # with a byte-identical kernel (10 015 -> 9 552 it/s round over round)
    device_rec, calib = None, None
    try:
        import ctypes as C_
        from pnp_admm_cnc_mri_amd import _lib as L_
        clk, cus_ = C_.c_int(0), C_.c_int(0)
        pci, arch = C_.create_string_buffer(32), C_.create_string_buffer(64)
        L_.check(L_.lib().pnp_device_info(local_rank, C_.byref(clk), C_.byref(cus_), pci, 32, arch, 64))
        device_rec = {'pci_bus_id': pci.value.decode(), 'arch': arch.value.decode(), 'clock_mhz': clk.value, 'compute_units': cus_.value}
        if args.size == 256 and args.precision == 'f32':
            gbs = C_.c_double(0.0)
            L_.check(L_.lib().pnp_calibrate_stream(local_rank, B, 0.6, C_.byref(gbs)))
            calib = {'calibration_gbs': gbs.value, 'seconds': 0.6, 'slices': B,
                     'kernel': 'k_calibrate_stream: the slice-resident loop\'s access shape without its arithmetic (one 512-thread workgroup per '
                               '256-KiB slice, 16 B per lane, 8 accesses in flight per wave; reads z, w, table, writes z, w)'}
    except Exception as e:                                           # attribution never breaks the line
        calib = {'error': repr(e)[:200]}

    def run(e, n):
        if args.solver == 'cnc':
            e.admm_cnc(n, PRESET['alpha'], PRESET['lambda1'], PRESET['reo'], PRESET['b'])
        else:
            e.admm_l1(n, 0.1, 0.015)

    def timed(e):
        """W untimed steps, then K timed ones.  Both ends of the timed region are a device sync on every rank with
        a job-wide barrier around them; a rank's clock stops at ITS OWN sync, before the closing barrier, so the
        collective's latency (tens to hundreds of microseconds, against a 2 ms region at the driver's 20 steps) is
        not booked as step time.  The job's time is the MAX over ranks (all_reduce further down)."""
        if args.warmup > 0:
            run(e, args.warmup)
        e.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.timer_start()
        run(e, args.steps)
        ev = e.timer_stop()                                           # HIP events on the kernels' stream; waits for them
        e.sync()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        if dist is not None:
            dist.barrier()
        return wall, ev

    wall_ms, ev_ms = timed(eng)

    # the one collective of the job: gather x on rank 0 over RCCL (timed apart from the steps)
    gather_ms = None
    x_dev = torch.empty((B, H, W), dtype=torch.float64 if args.precision == 'f64' else torch.float32, device='cuda')
    eng.x(out=x_dev)                                                  # x after exactly W + K iterations: what checksum / parity / gather see
    eng.sync()

    def sustained(e, seconds, ms_per_step_burst):
        """The job as configs[1] states it, back to back: { state <- z0 = |ifft2(y)|, w0 = 0;  ONE call of --sustain-steps
        (100) iterations } repeated for `seconds` s after a pre-heat of half that, no host sync in between; HIP events on
        the kernels' stream around the whole timed span.  Only iterations are counted; the two initialisation kernels
        and the state-order conversion of every solve are inside the span.  -> (steps, span_ms)."""
        ks = args.sustain_steps
        per_call_ms = max(ms_per_step_burst * ks * 1.1, 1e-3)
        n_heat = max(1, int(seconds * 0.5 * 1e3 / per_call_ms))
        n_timed = max(1, int(seconds * 1e3 / per_call_ms))

        def calls(n):
            for _ in range(n):
                if args.sustain_reinit:
                    e.init_state()
                run(e, ks)
        calls(n_heat)
        e.timer_start()
        calls(n_timed)
        return n_timed * ks, e.timer_stop()

    sus = None
    if args.sustain_s > 0:
        sus_steps, sus_ms = sustained(eng, args.sustain_s, ev_ms / args.steps)
        sus = [float(sus_steps), float(sus_ms)]

    # `l1` sub-record (N = 1, headline configuration): ADMM_L1 -- the solver of BASELINE.json configs[0], S1:171 presets -- on the
    # same resident batch: a burst of K steps after W warm-up steps from a fresh state, then the sustained form (half the span).
    l1_rec = None
    if (world == 1 and args.solver == 'cnc' and args.size == 256 and args.precision == 'f32' and not args.generic and not args.no_l1_record):
        def run_l1(e, n):
            e.admm_l1(n, 0.1, 0.015)
        eng.init_state()
        if args.warmup > 0:
            run_l1(eng, args.warmup)
        eng.sync()
        eng.timer_start()
        run_l1(eng, args.steps)
        l1_ms = eng.timer_stop()
        xl = torch.empty((B, H, W), dtype=torch.float32, device='cuda')
        eng.x(out=xl)
        eng.sync()
        l1_rec = {'burst_ms': float(l1_ms), 'path': eng.path_name, 'x': xl}
        if args.sustain_s > 0:
            ks = args.sustain_steps
            per_call = max(l1_ms / args.steps * ks * 1.1, 1e-3)
            n_heat, n_timed = max(1, int(args.sustain_s * 0.25 * 1e3 / per_call)), max(1, int(args.sustain_s * 0.5 * 1e3 / per_call))
            for _ in range(n_heat):
                eng.init_state(); run_l1(eng, ks)
            eng.timer_start()
            for _ in range(n_timed):
                eng.init_state(); run_l1(eng, ks)
            l1_rec['sus'] = (n_timed * ks, float(eng.timer_stop()))
    if dist is not None:
        from pnp_admm_cnc_mri_amd import sharding
        torch.cuda.synchronize()
        dist.barrier()
        xg = x_dev.cpu() if args.rehearse_gloo else x_dev
        sharding.gather_slices(xg, world * B, dst=0)                   # untimed: the first collective sets up the peer connections
        torch.cuda.synchronize()
        dist.barrier()
        tg = time.perf_counter()
        x_all = sharding.gather_slices(xg, world * B, dst=0)          # one direct RCCL gather over xGMI
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg) * 1e3
        assert (x_all is not None) == (rank == 0)
        if rank == 0:
            assert tuple(x_all.shape) == (world * B, H, W) and torch.equal(x_all[:B].to(x_dev.device), x_dev)
        dev_t = 'cpu' if args.rehearse_gloo else 'cuda'
        mine = torch.tensor([wall_ms, ev_ms, gather_ms, sus[1] / sus[0] if sus else 0.0], dtype=torch.float64, device=dev_t)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)                                    # every rank's own clocks: a slow GPU shows in the line
        per_rank = torch.stack(every).cpu()
        wall_ms, ev_ms, gather_ms = (float(v) for v in per_rank[:, :3].max(dim=0).values)   # the job's time: MAX over ranks
        if sus:
            sus = [sus[0], float(per_rank[:, 3].max()) * sus[0]]      # slowest rank's time per step x this rank's steps (equal on all ranks up to the burst estimate)
    per_rank_ms = None if dist is None else {'ms_per_step': [float(v) / args.steps for v in per_rank[:, 0]],
                                             'hip_event_ms_per_step': [float(v) / args.steps for v in per_rank[:, 1]],
                                             'gather_ms': [float(v) for v in per_rank[:, 2]],
                                             'sustained_ms_per_step': [float(v) for v in per_rank[:, 3]] if sus else None}
    checksum = float(x_dev.double().sum())
    finite = bool(torch.isfinite(x_dev).all())

    # Checker legs (rank 0 of an N = 1 job, with the CPU baseline): the NumPy oracle runs the same W + K iterations on
    # three of the slices from the same measurements, so every line re-proves parity of what was timed; and the
    # double-precision engine -- the configuration that holds 1e-5 against the float64 reference at config 2's own
    # 100 CNC iterations -- is timed in the same process on the same batch.
    parity, f64_record = None, None
    checker = rank == 0 and world == 1 and not args.no_cpu_baseline
    if checker:
        from oracle import admm_oracle as O
        n_it = args.warmup + args.steps
        picks = sorted({0, B // 2 - 1 if B > 1 else 0, B - 1})
        y_all = eng.download_y() if args.precision == 'f32' else None
        if args.precision == 'f64':                                    # the f64 engine holds y in double: rebuild the float32 measurements
            e32 = P.Engine(H, W, Bmax=B, device=local_rank)
            e32.synthesize(img, noise, masks, mask_id)
            y_all = e32.download_y()
            e32.close()
        ref = {}
        for b in picks:
            y64 = y_all[b].astype(np.complex128)
            ref[b] = (O.admm_cnc(y64, masks[mask_id[b]], n_it, **PRESET) if args.solver == 'cnc'
                      else O.admm_l1(y64, masks[mask_id[b]], n_it, lambda1=0.1, reo=0.015))

        def rel(xd):
            return [float(np.linalg.norm(xd[b].double().cpu().numpy() - ref[b]) / np.linalg.norm(ref[b])) for b in picks]
        parity = {'rel_l2_vs_oracle': rel(x_dev), 'slices': picks, 'iterations': n_it, 'oracle': 'oracle/admm_oracle.py (NumPy float64)'}
        if l1_rec is not None:
            l1_rec['rel'] = [float(np.linalg.norm(l1_rec['x'][b].double().cpu().numpy() - r) / np.linalg.norm(r))
                             for b, r in ((b, O.admm_l1(y_all[b].astype(np.complex128), masks[mask_id[b]], n_it, lambda1=0.1, reo=0.015)) for b in picks)]
        if (args.precision == 'f32' and args.size == 256 and args.solver == 'cnc' and not args.generic
                and not args.no_f64_record):
            e64 = P.Engine(H, W, Bmax=B, device=local_rank, precision='f64')
            e64.upload(y_all, masks, mask_id)
            e64.init_state()
            w64, ev64 = timed(e64)
            x64 = torch.empty((B, H, W), dtype=torch.float64, device='cuda')
            e64.x(out=x64)
            e64.sync()
            t64 = None
            try:
                t64 = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json'))).get(
                    '%s:cnc:256:f64:b%d' % (e64.path_name, B), {}).get('hbm_bytes_per_iteration')
            except Exception:
                pass
            f64_record = {'value': args.steps / (w64 * 1e-3) * (B / B_PER_GPU), 'unit': 'it/s (512-slice batches)',
                          'ms_per_step': w64 / args.steps, 'hip_event_ms_per_step': ev64 / args.steps,
                          'dtype': 'f64', 'path': e64.path_name, 'rel_l2_vs_oracle': rel(x64), 'slices': picks,
                          'iterations': n_it,
                          'frac': None if t64 is None else t64 / (ev64 / args.steps * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          'frac_contract_57N': ALG_BYTES_PER_PIXEL * H * W * B / (ev64 / args.steps * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          'traffic': t64}
            del x64
            e64.close()
        del y_all

    if rank == 0:
        K = args.steps
        ms_per_step = wall_ms / K
        value = world * K / (wall_ms * 1e-3) * (B / B_PER_GPU)
        alg_bytes = ALG_BYTES_PER_PIXEL * H * W * B * K
        # per-iteration HBM bytes of exactly this configuration from rocprofv3 --pmc (profiles/summarize.py)
        traffic, traffic_src = None, None
        tkey = '%s:%s:%d:%s:b%d' % (eng.path_name, args.solver, H, args.precision, B)
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath):
            try:
                ent = json.load(open(tpath)).get(tkey)
                if ent:
                    traffic, traffic_src = ent['hbm_bytes_per_iteration'], ent.get('from')
            except Exception:
                traffic = None
        sched = eng.schedule if eng.path_name == 'fused' else {'queues': 1, 'mixed': 0, 'chunk': 0}
        if args.generic or eng.path_name == 'generic':
            sched = {'queues': 1, 'mixed': 0, 'chunk': 0}
        ev_s_per_it = ev_ms * 1e-3 / K
        contract = alg_bytes / K                                            # 57 N B bytes per batched iteration
        own = OWN_BYTES_PER_PIXEL.get((eng.path_name, args.solver))
        own = None if own is None else own * H * W * B * (2 if args.precision == 'f64' else 1)
        # roofline.achieved / frac: the bytes the kernels REALLY move per iteration (PMC, committed per configuration) over the
        # HIP-event time -- a physical fraction of the 8 TB/s peak.  Where no PMC figure is committed for a configuration the
        # kernel family's own algorithmic bytes stand in (the PMC figure is 1.06x that on the headline path).
        phys_bytes, phys_kind = (traffic, 'pmc') if traffic is not None else (own, 'own_algorithmic') if own is not None else (contract, 'contract_57N')
        achieved = phys_bytes / ev_s_per_it / 1e9
        sustained_rec = None
        if sus:
            s_per_it = sus[1] * 1e-3 / sus[0]
            sustained_rec = {'value': world / s_per_it * (B / B_PER_GPU), 'unit': 'it/s (512-slice batches)',
                             'ms_per_step': s_per_it * 1e3, 'steps': int(sus[0]), 'steps_per_call': args.sustain_steps, 'span_s': sus[1] * 1e-3,
                             'preheat_s': args.sustain_s * 0.5, 'fresh_state_per_call': bool(args.sustain_reinit), 'frac': phys_bytes / s_per_it / 1e9 / HBM_PEAK_GBS,
                             'frac_contract_57N': contract / s_per_it / 1e9 / HBM_PEAK_GBS,
                             'timing': 'HIP events on the kernels\' stream around back-to-back solves { init_state; one call of '
                                       'steps_per_call iterations }, no host sync inside; initialisation inside the span, only iterations '
                                       'counted; N > 1: the slowest rank\'s time per step'}
        l1_line = None
        if l1_rec is not None:
            t_l1 = None
            try:
                t_l1 = json.load(open(tpath)).get('%s:l1:%d:f32:b%d' % (l1_rec['path'], H, B), {}).get('hbm_bytes_per_iteration')
            except Exception:
                pass
            own_l1 = OWN_BYTES_PER_PIXEL.get((l1_rec['path'], 'l1'))
            b_l1 = t_l1 if t_l1 is not None else (None if own_l1 is None else own_l1 * H * W * B)
            s_burst = l1_rec['burst_ms'] * 1e-3 / K
            l1_line = {'config': 'ADMM_L1 (S1:171 presets: lambda 0.1, reo 0.015), the same %d resident slices, %s path' % (B, l1_rec['path']),
                       'value': 1.0 / s_burst * (B / B_PER_GPU), 'unit': 'it/s (512-slice batches)', 'hip_event_ms_per_step': s_burst * 1e3,
                       'steps': K, 'warmup': args.warmup, 'bytes_per_iteration': b_l1, 'bytes_from': 'pmc' if t_l1 is not None else 'own_algorithmic',
                       'frac': None if b_l1 is None else b_l1 / s_burst / 1e9 / HBM_PEAK_GBS,
                       'frac_contract_57N': contract / s_burst / 1e9 / HBM_PEAK_GBS,
                       'rel_l2_vs_oracle': l1_rec.get('rel'), 'iterations': args.warmup + K}
            if 'sus' in l1_rec:
                s_sus = l1_rec['sus'][1] * 1e-3 / l1_rec['sus'][0]
                l1_line['sustained'] = {'value': 1.0 / s_sus * (B / B_PER_GPU), 'ms_per_step': s_sus * 1e3, 'steps': l1_rec['sus'][0],
                                        'span_s': l1_rec['sus'][1] * 1e-3, 'frac': None if b_l1 is None else b_l1 / s_sus / 1e9 / HBM_PEAK_GBS}
        line = {
            'metric': 'ADMM iterations/sec on %dx%d complex64 slices (batch=512)' % (H, W),
            'value': value, 'unit': 'it/s (512-slice batches)',
            'n_gpus': world, 'steps': K, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.precision, 'data': 'synthetic',
            'config': {'workload': 'ADMM_%s, %d synthetic %dx%d complex64 slices per GPU, %s, %s presets'
                                   % (args.solver.upper(), B, H, W, 'Q_Random30' if H == 256 else 'seeded mask bank of 3',
                                      'S4:176' if args.solver == 'cnc' else 'S1:171'),
                       'slices_per_gpu': B, 'path': eng.path_name, 'precision': args.precision,
                       # 0 = the iterations of a call are a loop inside ONE launch (slice-resident kernel)
                       'launches_per_iteration': eng.kernels_per_iteration,
                       'queues': eng.plan['queues'], 'slices_per_chunk': eng.plan['chunk'],
                       'mixed_row_col_launches': bool(sched['mixed']), 'device': device_rec},
            'slice_iterations_per_s': value * B_PER_GPU,
            'hip_event_ms_per_step': ev_ms / K,
            'sustained': sustained_rec,
            'per_rank': per_rank_ms,
            'gather_ms': gather_ms, 'x_checksum': checksum, 'x_finite': finite,
            'parity': parity, 'f64': f64_record, 'l1': l1_line,
            'roofline': {'schema': 2,       # 2 (round 4 on): achieved / frac are PHYSICAL (PMC bytes); rounds 1-3 priced the 57 N contract bytes under the same keys
                         'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'frac_physical': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'calibration': calib,
                         'frac_of_calibration': (achieved / calib['calibration_gbs']) if calib and calib.get('calibration_gbs') else None,
                         'bytes_per_iteration': phys_bytes, 'bytes_from': phys_kind,
                         'frac_own_algorithmic': None if own is None else own / ev_s_per_it / 1e9 / HBM_PEAK_GBS,
                         'own_algorithmic_bytes_per_iteration': own,
                         'frac_contract_57N': contract / ev_s_per_it / 1e9 / HBM_PEAK_GBS,
                         'achieved_contract_57N': contract / ev_s_per_it / 1e9,
                         'contract_bytes_per_iteration': contract,
                         'traffic_over_contract': None if traffic is None else traffic / contract,
                         'traffic_over_own_algorithmic': None if (traffic is None or own is None) else traffic / own,
                         'traffic_from': traffic_src,
                         'traffic_measured_in_this_run': False,       # PMC passes on the builder's box, committed under profiles/
                         'note': 'achieved / frac = HBM bytes the kernels move per batched iteration (traffic: rocprofv3 PMC, '
                                 'FETCH_SIZE x2 + WRITE_SIZE, of exactly this configuration) / HIP-event time per iteration on the '
                                 'kernels\' stream.  frac_contract_57N prices the same time against SURVEY.md 8(d)\'s 57*H*W*B bytes '
                                 'of the plain c2c float32 formulation; it exceeds 1 because the kernels move fewer bytes than that '
                                 '(slice-resident path: z, w and the Hermitian half-plane table only, the transposed field never '
                                 'leaves the compute unit; fused path: two real slices per complex FFT, Hermitian half plane).'},
        }
        if world == 1 and not args.no_cpu_baseline and args.size == 256 and args.precision == 'f32' and not (args.no_pnp_record and args.no_latency_record):
            # the sub-records get the whole card: this process lets go of its engine, its buffers and torch's cached blocks first
            eng.close()
            del x_dev
            l1_rec = None
            torch.cuda.empty_cache()
            if not args.no_latency_record:
                try:
                    line['latency'] = latency_record(local_rank)
                except Exception as e:                               # a sub-record never breaks the line
                    line['latency'] = {'error': repr(e)[:300]}
                torch.cuda.empty_cache()
            if not args.no_pnp_record:
                line['pnp'] = pnp_record()
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(masks, mask_id, args.cpu_budget)
            try:
                line['cpu_baseline_all_cores'] = cpu_baseline_all_cores(min(8.0, args.cpu_budget))
            except Exception as e:                                   # never let the extra leg break the line
                line['cpu_baseline_all_cores'] = {'error': repr(e)}
        else:
            line['cpu_baseline'] = None
        out.write(json.dumps(line) + '\n')
        out.flush()
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
