"""Two ranks (gloo rendezvous at 127.0.0.1, both on cuda:0 -- the test box has one GPU) run the real
ADMM_CNC entry point on their slice blocks through `sharding.solve_sharded`; the gathered result
must equal the single-process result bit for bit (slices are independent)."""
import functools
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problem():
    from pnp_admm_cnc_mri_amd import synthetic as S
    m = S.reference_masks()
    masks = np.stack([m['Q_Random30'], m['Q_Radial30'], m['Q_Cartesian30']]).astype(np.uint8)
    B = 7
    img, noise = S.batch(100, B)
    mid = (np.arange(B) % 3).astype(np.int32)
    y = np.stack([np.fft.fft2(img[b]) * masks[mid[b]] + noise[b] for b in range(B)]).astype(np.complex64)
    return masks, y, mid


def _worker(rank, world, port, q, precision='f32'):
    import torch.distributed as dist
    import pnp_admm_cnc_mri_amd as P
    from pnp_admm_cnc_mri_amd import sharding
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        masks, y, mid = _problem()
        solver = functools.partial(P.ADMM_CNC, device=0, results='/tmp/pnp_sharded_results_%d' % rank)
        x = sharding.solve_sharded(solver, masks, None, y=y, mask_id=mid, precision=precision, **P.PRESETS['ADMM_CNC'])
        if rank == 0:
            q.put(x)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('precision', ['f32', 'f64'])
def test_two_ranks_equal_one_process(precision):
    import torch.multiprocessing as mp
    import pnp_admm_cnc_mri_amd as P
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, precision)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    masks, y, mid = _problem()
    out = P.ADMM_CNC(masks, None, y=y, mask_id=mid, results='/tmp/pnp_sharded_results_ref', precision=precision, **P.PRESETS['ADMM_CNC'])
    real = np.float64 if precision == 'f64' else np.float32             # the gather keeps the solver's precision
    ref = np.stack([out[b].astype(real) for b in range(len(y))])
    assert got.dtype == real
    # rank 0 holds slices 0..3 (pairs (0,1),(2,3)), rank 1 slices 4..6; in the single process slice 4
    # is paired with 5 and 6 is alone in both cases, so pairing is identical -> bit-identical
    assert got.shape == ref.shape and np.array_equal(got, ref)


def test_return_device_hands_back_the_reconstructions_as_one_device_tensor(tmp_path):
    """`return_device=True` (what solve_sharded asks of every entry point before the RCCL gather): `out` is ONE [B,H,W] tensor on the
    device, equal to the host list of the plain call -- ADMM_CNC in both precisions, ADMM_L1, and the PnP entry points."""
    import torch
    import pnp_admm_cnc_mri_amd as P
    from pnp_admm_cnc_mri_amd import solvers_pnp as SP, denoisers as D
    masks, y, mid = _problem()
    kw = dict(y=y, mask_id=mid, results=str(tmp_path))
    for fn, opts in ((P.ADMM_CNC, dict(P.PRESETS['ADMM_CNC'], iter_num=4)), (P.ADMM_CNC, dict(P.PRESETS['ADMM_CNC'], iter_num=4, precision='f64')),
                     (P.ADMM_L1, dict(P.PRESETS['ADMM_L1'], iter_num=4))):
        host = fn(masks, None, **kw, **opts)
        dev = fn(masks, None, return_device=True, **kw, **opts)
        assert torch.is_tensor(dev) and dev.is_cuda and tuple(dev.shape) == (7, 256, 256)
        assert dev.dtype == (torch.float64 if opts.get('precision') == 'f64' else torch.float32)
        assert np.array_equal(dev.cpu().numpy().astype(np.float64), np.stack([host[b] for b in range(7)]))
    sd = D.seeded_state_dict(D.build('ffdnet_gray')[0], 1)
    o = dict(alpha=0.9, iter_num=2, lambda1=1.35, reo=0.45, b=0.3)
    host, psnr_h = SP.PNP_ADMM_CNC_D('ffdnet_gray', masks, None, model=sd, **kw, **o)
    dev, psnr_d = SP.PNP_ADMM_CNC_D('ffdnet_gray', masks, None, model=sd, return_device=True, **kw, **o)
    assert torch.is_tensor(dev) and dev.is_cuda and np.array_equal(dev.cpu().numpy().astype(np.float64), np.stack([host[b] for b in range(7)]))
    host = SP.PNP_ADMM_L1_D('ffdnet_gray', masks, None, model=sd, iter_num=2, reo=0.25, **kw)
    dev = SP.PNP_ADMM_L1_D('ffdnet_gray', masks, None, model=sd, iter_num=2, reo=0.25, return_device=True, **kw)
    assert torch.is_tensor(dev) and np.array_equal(dev.cpu().numpy().astype(np.float64), np.stack([host[b] for b in range(7)]))


def _worker_rccl(port, q):
    """one rank on RCCL: the real backend's first contact with solve_sharded (a child process: the suite's own process must not keep
    a process group)"""
    import torch
    import torch.distributed as dist
    import pnp_admm_cnc_mri_amd as P
    from pnp_admm_cnc_mri_amd import sharding
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), LOCAL_RANK='0')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        masks, y, mid = _problem()
        seen = {}
        real_gather = sharding.gather_slices

        def spy(x_local, B_total, dst=0, group=None):
            seen['device'], seen['dtype'] = x_local.device, x_local.dtype
            return real_gather(x_local, B_total, dst=dst, group=group)
        sharding.gather_slices = spy
        # the documented way for a bound solver: device in the partial, NOT in solve_sharded's keywords
        bound = functools.partial(P.ADMM_CNC, device=0, results='/tmp/pnp_sharded_results_rccl')
        x = sharding.solve_sharded(bound, masks, None, y=y, mask_id=mid, **P.PRESETS['ADMM_CNC'])
        # and the default: no device anywhere -> LOCAL_RANK under the process group
        x_default = sharding.solve_sharded(functools.partial(P.ADMM_CNC, results='/tmp/pnp_sharded_results_rccl'), masks, None, y=y,
                                           mask_id=mid, **P.PRESETS['ADMM_CNC'])
        q.put((x, x_default, str(seen['device']), str(seen['dtype'])))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_solve_sharded_through_rccl_with_one_rank():
    """`solve_sharded` on the 'nccl' (= RCCL) backend, one rank, the solver's device bound with functools.partial: the gather is handed
    the tensor on the rank's own card and the result is bit-equal to the plain call."""
    import torch.multiprocessing as mp
    import pnp_admm_cnc_mri_amd as P
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_worker_rccl, args=(port, q))
    p.start()
    x, x_default, dev, dt = q.get(timeout=300)
    p.join(timeout=300)
    assert p.exitcode == 0
    assert dev == 'cuda:0' and dt == 'torch.float32'
    masks, y, mid = _problem()
    out = P.ADMM_CNC(masks, None, y=y, mask_id=mid, results='/tmp/pnp_sharded_results_ref', **P.PRESETS['ADMM_CNC'])
    ref = np.stack([out[b].astype(np.float32) for b in range(len(y))])
    assert np.array_equal(x, ref) and np.array_equal(x_default, ref)
