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
