"""Multi-process path on CPU: world_size 2 (and 3, uneven) over gloo at 127.0.0.1.
Slices shard with no data-path collective; one gather ends the job; the gathered result must
equal the single-process result bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pnp_admm_cnc_mri_amd import sharding
from oracle import admm_oracle as O


def test_shard_ranges_partition():
    for B in (1, 2, 5, 512, 513, 4096):
        for world in (1, 2, 3, 8):
            r = [sharding.shard_range(B, world, k) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == B
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1 and sizes == sharding.shard_sizes(B, world)
    with pytest.raises(ValueError):
        sharding.shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _solve(lo, hi):
    """stand-in for the device loop on CPU: two oracle L1 iterations per slice (slice-independent)."""
    mask = (np.arange(64 * 64).reshape(64, 64) % 3 == 0).astype(np.float64)
    mask[0, 0] = 1
    out = []
    for b in range(lo, hi):
        rng = np.random.default_rng(b)
        y = np.fft.fft2(rng.uniform(0, 1, (64, 64))) * mask
        out.append(O.admm_l1(y, mask, 2).astype(np.float32))
    return torch.from_numpy(np.stack(out)) if out else torch.empty((0, 64, 64))


def _worker(rank, world, port, B, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        x = sharding.run_sharded(_solve, B)
        if rank == 0:
            q.put(x.numpy())
        else:
            assert x is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,B', [(2, 6), (2, 5), (3, 7)])
def test_sharded_equals_single_process(world, B):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref = _solve(0, B).numpy()
    assert got.shape == ref.shape and np.array_equal(got, ref)


def _fake_solver(mask, noises, images=None, y=None, mask_id=None, return_device=False, **opts):
    """same call shape as ADMM_L1: returns the reference-style list (22 slots) of float64 arrays -- or, asked with return_device=True
    as solve_sharded does, ONE [b,H,W] float32 tensor (here on the CPU: the gloo stand-in of the device tensor)."""
    out = [np.zeros(mask.shape[-2:], np.uint8)] * max(22, len(y))
    for n in range(len(y)):
        k = 0 if mask_id is None else int(mask_id[n])
        out[n] = O.admm_l1(y[n], mask[k], opts.get('iter_num', 2))
    if return_device:
        return torch.from_numpy(np.stack([np.asarray(out[n], np.float32) for n in range(len(y))]))
    return out


def _worker_solver(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        x = sharding.solve_sharded(_fake_solver, *_solver_problem()[:2], y=_solver_problem()[2], mask_id=_solver_problem()[3], iter_num=2)
        if rank == 0:
            q.put(x)
        else:
            assert x is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _solver_problem():
    rng = np.random.default_rng(0)
    masks = (rng.uniform(size=(2, 64, 64)) < 0.4).astype(np.float64)
    masks[:, 0, 0] = 1
    B = 5
    mid = np.arange(B) % 2
    y = np.stack([np.fft.fft2(np.random.default_rng(b).uniform(0, 1, (64, 64))) * masks[mid[b]] for b in range(B)])
    return masks, None, y, mid


def test_solve_sharded_entry_point_shape():
    """An entry-point-shaped solver sharded over 2 ranks (uneven: 3 + 2 slices, per-slice masks)
    equals the single-process call."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_solver, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    masks, _, y, mid = _solver_problem()
    ref = sharding.solve_sharded(_fake_solver, masks, None, y=y, mask_id=mid, iter_num=2)
    assert got.shape == (5, 64, 64) and np.array_equal(got, ref)


def test_run_sharded_without_process_group():
    x = sharding.run_sharded(_solve, 3)
    assert x.shape == (3, 64, 64)


def _noisy_solver(mask, noises, images=None, y=None, mask_id=None, **opts):
    """entry-point shape with `images` + per-slice k-space noise [b,H,W]: synthesises y like S4:102,
    and (like Engine.synthesize) refuses a noise batch that does not match the images."""
    noises = np.asarray(noises)
    if noises.ndim == 3 and noises.shape[0] != len(images):
        raise ValueError('noise batch does not match images')
    out = [np.zeros(mask.shape[-2:], np.uint8)] * max(22, len(images))
    for n in range(len(images)):
        nz = noises[n] if noises.ndim == 3 else noises
        out[n] = O.admm_l1(O.synthesize(images[n], mask, nz), mask, 2)
    return out


def _noisy_problem(B=5):
    rng = np.random.default_rng(3)
    mask = (rng.uniform(size=(64, 64)) < 0.4).astype(np.float64)
    mask[0, 0] = 1
    imgs = rng.uniform(0, 1, (B, 64, 64)).astype(np.float32)
    noises = (rng.standard_normal((B, 64, 64)) + 1j * rng.standard_normal((B, 64, 64))) * 0.5
    return mask, noises, imgs


def _worker_noisy(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        mask, noises, imgs = _noisy_problem()
        # (1) per-slice noise, uneven shards (2 + 2 + 1), gathered on global rank 0
        x = sharding.solve_sharded(_noisy_solver, mask, noises, images=imgs)
        assert (x is not None) == (rank == 0)
        # (2) a sub-group {1, 2} gathering on GLOBAL rank 2 (group-local rank 1)
        grp = dist.new_group([1, 2])
        x2 = None
        if rank in (1, 2):
            x2 = sharding.solve_sharded(_noisy_solver, mask, noises, images=imgs, dst=2, group=grp)
            assert (x2 is not None) == (rank == 2)
        if rank == 0:
            q.put(('all', x))
        if rank == 2:
            q.put(('sub', x2))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_solve_sharded_per_slice_noise_and_subgroup_dst():
    """per-slice noise [B,H,W] is sharded with the slices (uneven shards over 3 ranks), and `dst` is a
    global rank also when the job runs in a sub-group."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_noisy, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    mask, noises, imgs = _noisy_problem()
    ref = sharding.solve_sharded(_noisy_solver, mask, noises, images=imgs)
    assert got['all'].shape == (5, 64, 64) and np.array_equal(got['all'], ref)
    assert np.array_equal(got['sub'], ref)


# ----------------------------------------------------------------------------------------------
# First contact with 8 GPUs, rehearsed on the CPU: world 8 over gloo with the shard arithmetic of BASELINE.json configs[3] / [4]
# (4096 / 8 and 2048 / 8 slices per rank at 256^2 / 512^2 -- here the same counts scaled down to a tiny H x W so that 8 processes
# finish in seconds), per-slice mask ids b % 3, and an uneven remainder case.  Bit-equal to one process.
# ----------------------------------------------------------------------------------------------
def _tiny_solver(mask, noises, images=None, y=None, mask_id=None, return_device=False, **opts):
    """entry-point shape on 8 x 8 slices: one DC step + soft threshold per slice (slice-independent, mask bank + mask_id)"""
    xs = []
    for n in range(len(y)):
        m = mask[int(mask_id[n])]
        X = np.fft.fft2(np.abs(np.fft.ifft2(y[n])))
        X[m > 0] = (2.0 * X[m > 0] + y[n][m > 0]) / 3.0
        x = np.abs(np.real(np.fft.ifft2(X)))
        xs.append(np.float32(np.sign(x) * np.maximum(np.abs(x) - 0.01, 0)))
    t = torch.from_numpy(np.stack(xs))
    return t if return_device else [a.astype(np.float64) for a in xs]


def _tiny_problem(B):
    rng = np.random.default_rng(8)
    masks = (rng.uniform(size=(3, 8, 8)) < 0.4).astype(np.float64)
    masks[:, 0, 0] = 1
    mid = (np.arange(B) % 3).astype(np.int32)                        # config 5: mask_id = b % 3
    y = (rng.standard_normal((B, 8, 8)) + 1j * rng.standard_normal((B, 8, 8))) * masks[mid]
    return masks, y, mid


def _worker8(rank, world, port, Bs, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        for B in Bs:
            masks, y, mid = _tiny_problem(B)
            lo, hi = sharding.shard_range(B, world, rank)
            x = sharding.solve_sharded(_tiny_solver, masks, None, y=y, mask_id=mid)
            assert (x is not None) == (rank == 0)
            if rank == 0:
                q.put((B, x))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_world_eight_with_the_shard_counts_of_configs_4_and_5():
    Bs = (4096, 2048, 2051, 5)             # 512 and 256 slices per rank; 2051: three ranks hold one slice more; 5: three ranks hold none
    assert sharding.shard_sizes(4096, 8) == [512] * 8 and sharding.shard_sizes(2048, 8) == [256] * 8
    assert sharding.shard_sizes(2051, 8) == [257] * 3 + [256] * 5 and sharding.shard_sizes(5, 8) == [1] * 5 + [0] * 3
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, Bs, q)) for r in range(8)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in Bs)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for B in Bs:
        masks, y, mid = _tiny_problem(B)
        ref = sharding.solve_sharded(_tiny_solver, masks, None, y=y, mask_id=mid)
        assert got[B].shape == (B, 8, 8) and got[B].dtype == np.float32 and np.array_equal(got[B], ref), B


# ---- round 6: the collective's device and the entry points' default device (first contact with 8 GPUs) ----
class _OnCard:
    """what collective_device looks at of a tensor: where it lies"""
    def __init__(self, index):
        self.is_cuda, self.device = index is not None, (torch.device('cuda', index) if index is not None else torch.device('cpu'))


def test_the_collective_runs_on_the_card_the_result_lies_on(monkeypatch):
    """A solver bound to its GPU with functools.partial(..., device=3) never shows `device` to solve_sharded: RCCL must get the tensor
    on cuda:3 where it already is -- not a copy onto cuda:0, the root's card (round 5: `kw.get('device', 0)`)."""
    assert sharding.collective_device(_OnCard(3), 'nccl') == torch.device('cuda', 3)
    assert sharding.collective_device(_OnCard(3), 'nccl', device=0) == torch.device('cuda', 3)      # where the data IS wins
    assert sharding.collective_device(_OnCard(5), 'gloo') == torch.device('cpu')
    # a CPU tensor under nccl (an empty shard; a list-returning solver): explicit device=, else the process's LOCAL_RANK
    assert sharding.collective_device(_OnCard(None), 'nccl', device=6) == torch.device('cuda', 6)
    monkeypatch.setenv('LOCAL_RANK', '5')
    monkeypatch.setattr(dist, 'is_initialized', lambda: True)
    assert sharding.collective_device(_OnCard(None), 'nccl') == torch.device('cuda', 5)


def test_the_default_device_is_the_local_rank_under_a_process_group(monkeypatch):
    from pnp_admm_cnc_mri_amd.solvers import resolve_device
    monkeypatch.setenv('LOCAL_RANK', '3')
    assert resolve_device(None) == 0                      # no process group: the reference's single-process usage
    assert resolve_device(2) == 2 and resolve_device(torch.device('cuda', 4)) == 4
    monkeypatch.setattr(dist, 'is_initialized', lambda: True)
    assert resolve_device(None) == 3                      # one process per GPU: rank r works on its own card
    assert resolve_device(1) == 1                         # an explicit device= still wins


def _strict_solver(mask, noises, y=None, mask_id=None, iter_num=2):
    """no return_device keyword, no **opts, float64 list out: the reference's own call shape"""
    return [O.admm_l1(y[n], mask[int(mask_id[n])], iter_num) for n in range(len(y))]


def test_solve_sharded_passes_the_partial_solvers_tensor_to_the_collective(monkeypatch, tmp_path):
    """solve_sharded under a (one-rank, gloo) process group: the collective runs -- a one-rank group is not short-circuited -- on the
    device collective_device picks from the solver's OWN result, and a solver without the `return_device` keyword is called without
    it, its float64 list cast to the gather's dtype (advisor, round 5)."""
    import functools
    seen = {}
    real_cd = sharding.collective_device

    def spy(x_local, backend, device=None):
        seen['x'], seen['backend'], seen['device_kw'] = x_local, backend, device
        return real_cd(x_local, backend, device)
    monkeypatch.setattr(sharding, 'collective_device', spy)
    dist.init_process_group('gloo', rank=0, world_size=1, init_method='file://%s' % (tmp_path / 'rdv'))
    try:
        masks, _, y, mid = _solver_problem()
        got = sharding.solve_sharded(functools.partial(_fake_solver, iter_num=2), masks, None, y=y, mask_id=mid)
        assert seen['backend'] == 'gloo' and seen['device_kw'] is None and torch.is_tensor(seen['x']) and seen['x'].shape[0] == 5
        strict = sharding.solve_sharded(_strict_solver, masks, None, y=y, mask_id=mid, iter_num=2)
    finally:
        dist.destroy_process_group()
    ref = sharding.solve_sharded(_fake_solver, masks, None, y=y, mask_id=mid, iter_num=2)        # no group: a plain call
    assert np.array_equal(got, ref)
    assert strict.dtype == np.float32 and np.array_equal(strict, ref)
