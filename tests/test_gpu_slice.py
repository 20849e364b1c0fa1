"""GPU parity of the slice-resident 256x256 loops (kernels_slice256.hip): one workgroup keeps one slice
in the register file for the whole run; only z, w and the Hermitian table travel to HBM."""
import numpy as np
import pytest

from oracle import admm_oracle as O
from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def P():
    import pnp_admm_cnc_mri_amd as P
    from pnp_admm_cnc_mri_amd import _lib
    assert _lib.device_count() >= 1
    return P


def _problem(golden_inputs, B):
    masks = np.stack([golden_inputs['masks'][k] for k in ('Q_Random30', 'Q_Radial30', 'Q_Cartesian30')]).astype(np.uint8)
    mid = (np.arange(B) % 3).astype(np.int32)
    ys = np.stack([O.synthetic_problem(b, masks[mid[b]])[1] for b in range(B)]).astype(np.complex64)
    return masks, mid, ys


@pytest.mark.parametrize('iters', [1, 2, 7])
@pytest.mark.parametrize('solver', ['cnc', 'l1', 'l1_two_state'])
def test_slice_resident_loops_vs_oracle(P, golden_inputs, solver, iters, monkeypatch):
    B = 5
    masks, mid, ys = _problem(golden_inputs, B)
    monkeypatch.setenv('PNP_SLICE', '1')
    if solver == 'l1_two_state':
        monkeypatch.setenv('PNP_FUSED_L1_TWO_STATE', '1')
    with P.Engine(256, 256, Bmax=B) as eng:
        eng.upload(ys, masks, mid)
        assert eng.path_name == 'slice'
        eng.init_state()
        if solver == 'cnc':
            eng.admm_cnc(iters, 0.45, 0.5, 0.05, 64)
        else:
            eng.admm_l1(iters, 0.1, 0.015)
        x = eng.x()
        z, w = eng.get_state()
    for b in range(B):
        y128 = ys[b].astype(np.complex128)
        if solver == 'cnc':
            ref = O.admm_cnc(y128, masks[mid[b]], iters)
        else:
            ref = O.admm_l1(y128, masks[mid[b]], iters)
        assert rel_l2(x[b], ref) <= 2e-6, (b, rel_l2(x[b], ref))
    assert np.isfinite(z).all() and np.isfinite(w).all()


def test_slice_resident_equals_two_launch_path_state(P, golden_inputs, monkeypatch):
    """x, z AND w after a run agree with the two-launch fused path (same cores, different data flow),
    and a run split into two calls is bit-identical to one call."""
    B = 4
    masks, mid, ys = _problem(golden_inputs, B)
    res = {}
    for slice_on in ('0', '1'):
        monkeypatch.setenv('PNP_SLICE', slice_on)
        with P.Engine(256, 256, Bmax=B) as eng:
            eng.upload(ys, masks, mid)
            eng.init_state()
            eng.admm_cnc(6, 0.45, 0.5, 0.05, 64)
            res[slice_on] = (eng.x(), *eng.get_state())
            if slice_on == '1':
                eng.init_state()
                eng.admm_cnc(2, 0.45, 0.5, 0.05, 64)
                eng.admm_cnc(4, 0.45, 0.5, 0.05, 64)
                assert np.array_equal(eng.x(), res['1'][0])
    for a, b in zip(res['0'], res['1']):
        assert np.abs(a - b).max() <= 1e-5          # two data flows, 6 CNC iterations of rounding growth (measured 3.6e-6)


def test_path_selection_by_batch_size(P, monkeypatch):
    """Which loops run slice-resident (pnp_path_name): batches of at least 64 slices (PNP_SLICE_MIN_B); PNP_SLICE=0 / 1
    override."""
    monkeypatch.delenv('PNP_SLICE', raising=False)
    mask = np.ones((256, 256), np.uint8)
    with P.Engine(256, 256, Bmax=520) as eng:
        for B, want in ((6, 'fused'), (63, 'fused'), (64, 'slice'), (131, 'slice'), (256, 'slice'), (288, 'slice'),
                        (512, 'slice'), (520, 'slice')):
            eng.upload(np.zeros((B, 256, 256), np.complex64), mask)
            assert eng.path_name == want, (B, eng.path_name)
    with P.Engine(256, 256, Bmax=48) as eng:             # a context that can never hold a slice-resident batch has no tables
        eng.upload(np.zeros((48, 256, 256), np.complex64), mask)
        assert eng.path_name == 'fused'
    monkeypatch.setenv('PNP_SLICE', '0')
    with P.Engine(256, 256, Bmax=512) as eng:
        eng.upload(np.zeros((512, 256, 256), np.complex64), mask)
        assert eng.path_name == 'fused'


def test_slice_resident_runs_are_bitwise_repeatable(P, monkeypatch):
    """The kernel hands data between lanes and waves through LDS (exchanges, transpositions, one LDS-DMA prefetch for the
    packed column) under barriers and wait counts placed by hand or by the compiler: a missing wait shows as results
    that change from run to run.  40 identical runs of a two-round batch (300 slices > 256 compute units) with different
    masks per slice must agree to the bit, for x and for the state."""
    monkeypatch.setenv('PNP_SLICE', '1')
    from pnp_admm_cnc_mri_amd import synthetic as S
    m = S.reference_masks()
    masks = np.stack([m['Q_Random30'], m['Q_Cartesian30'], m['Q_Radial30']]).astype(np.uint8)
    B = 300
    img, noise = S.batch(7, B)
    with P.Engine(256, 256, Bmax=B) as eng:
        eng.synthesize(img, noise, masks, np.arange(B) % 3)
        assert eng.path_name == 'slice'
        ref = None
        for rep in range(40):
            eng.init_state()
            if rep % 2:
                eng.admm_cnc(5, 0.45, 0.5, 0.05, 64)
            else:                                   # the same 5 iterations as two launches
                eng.admm_cnc(2, 0.45, 0.5, 0.05, 64)
                eng.admm_cnc(3, 0.45, 0.5, 0.05, 64)
            got = (eng.x(), *eng.get_state())
            if ref is None:
                ref = got
            else:
                for a, b in zip(ref, got):
                    assert np.array_equal(a, b), rep


@pytest.mark.parametrize('B', [1, 2, 63, 64, 65, 255, 256, 257, 300])
def test_slice_path_at_every_batch_shape(P, B, monkeypatch):
    """Batch sizes around the 64-slice selection threshold and around one round of 256 compute units (a second round
    of one slice, of 44): the slice-resident loops against the two-launch path for x, z, w (values in [0, 1]:
    max-abs <= 2e-5; measured 6.2e-6) and against the oracle on the first, middle and last slice."""
    from pnp_admm_cnc_mri_amd import synthetic as S
    m = S.reference_masks()
    masks = np.stack([m['Q_Random30'], m['Q_Cartesian30'], m['Q_Radial30']]).astype(np.uint8)
    mid = (np.arange(B) % 3).astype(np.int32)
    img, noise = S.batch(0, B)
    res = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('PNP_SLICE', mode)
        with P.Engine(256, 256, Bmax=B) as eng:
            eng.synthesize(img, noise, masks, mid)
            assert eng.path_name == ('slice' if mode == '1' else 'fused')
            y = eng.download_y()
            eng.init_state()
            eng.admm_cnc(6, 0.45, 0.5, 0.05, 64)
            x = eng.x()
            z, w = eng.get_state()
            eng.init_state()
            eng.admm_l1(5, 0.1, 0.015)
            res[mode] = (x, z, w, eng.x())
    for a, b in zip(res['1'], res['0']):
        assert np.abs(a.astype(np.float64) - b.astype(np.float64)).max() <= 2e-5
    for b in sorted({0, B // 2, B - 1}):
        y128 = y[b].astype(np.complex128)
        assert rel_l2(res['1'][0][b], O.admm_cnc(y128, masks[mid[b]], 6)) <= 2e-6
        assert rel_l2(res['1'][3][b], O.admm_l1(y128, masks[mid[b]], 5)) <= 2e-6


def test_padded_and_unpadded_slice_strides_are_bit_identical(P, monkeypatch):
    """The slice path keeps its state and its table with 4 KiB of padding per slice (PNP_SLICE_PAD_KB / PNP_SLICE_YH_PAD_KB,
    kernels_slice256.hip: slice256_create); 0 is the in-place, unpadded form.  Same arithmetic on the same values: x, z and w
    agree to the bit -- across a state hand-over (get_state / set_state go through the natural order) and a batch that
    ends in a partial round."""
    from pnp_admm_cnc_mri_amd import synthetic as S
    B = 259
    masks = S.reference_masks()['Q_Random30'].astype(np.uint8)[None]
    img, noise = S.batch(3, B)
    monkeypatch.setenv('PNP_SLICE', '1')
    res = {}
    for pad, yh in (('0', '0'), ('4', '4'), ('1', '16')):
        monkeypatch.setenv('PNP_SLICE_PAD_KB', pad)
        monkeypatch.setenv('PNP_SLICE_YH_PAD_KB', yh)
        with P.Engine(256, 256, Bmax=B) as eng:
            eng.synthesize(img, noise, masks, np.zeros(B, np.int32))
            eng.init_state()
            eng.admm_cnc(3, 0.45, 0.5, 0.05, 64)
            z, w = eng.get_state()                      # slice order -> natural
            eng.set_state(z, w)                         # natural; the next run converts again
            eng.admm_cnc(4, 0.45, 0.5, 0.05, 64)
            assert eng.path_name == 'slice'
            res[(pad, yh)] = (eng.x(),) + tuple(eng.get_state())
    ref = res[('0', '0')]
    for key, got in res.items():
        for a, b in zip(ref, got):
            assert np.array_equal(a, b), key


def test_chip_filling_run_matches_the_two_launch_path(P, monkeypatch):
    """512 slices (two rounds on every compute unit, the shape bench.py times), 48 iterations of the slice-resident
    kernel, EVERY slice checked against the two-launch path every 4 iterations.  Both paths share the arithmetic cores
    but not the data flow, so a store that lands wrong anywhere in the batch (buffer-store hazard) shows.
    The CNC map's steepest slope is 1.27 (1.27^50 = 1.5e5: end-to-end comparisons of 50 iterations drown a wrong value in
    amplified round-off), so the comparison is teacher-forced: every 4 iterations the two-launch engine restarts from the
    slice engine's state, and x, z, w after the next 4 must agree per pixel to 2e-5 (values in [0, 1]; the two data flows
    differ by ~3e-7 per pixel and iteration; a misplaced store is O(0.01 .. 0.5))."""
    from pnp_admm_cnc_mri_amd import synthetic as S
    m = S.reference_masks()
    masks = np.stack([m['Q_Random30']]).astype(np.uint8)
    B = 512
    img, noise = S.batch(0, B)
    engs = {}
    try:
        for mode in ('1', '0'):
            monkeypatch.setenv('PNP_SLICE', mode)
            engs[mode] = P.Engine(256, 256, Bmax=B)
            engs[mode].synthesize(img, noise, masks, np.zeros(B, np.int32))
            assert engs[mode].path_name == ('slice' if mode == '1' else 'fused')
        engs['1'].init_state()
        worst = 0.0
        for chunk in range(12):
            z, w = engs['1'].get_state()
            engs['0'].set_state(z, w)
            res = {}
            for mode in ('1', '0'):
                engs[mode].admm_cnc(4, 0.45, 0.5, 0.05, 64)
                res[mode] = (engs[mode].x(), *engs[mode].get_state())
            for name, a, b in zip('xzw', res['1'], res['0']):
                d = np.abs(a.astype(np.float64) - b.astype(np.float64)).reshape(B, -1).max(axis=1)
                worst = max(worst, float(d.max()))
                assert d.max() <= 2e-5, (chunk, name, int(d.argmax()), float(d.max()))
        assert worst > 0                       # two different data flows: not the same kernel compared with itself
    finally:
        for e in engs.values():
            e.close()


def test_state_order_is_private_to_the_slice_path(P, golden_inputs, monkeypatch):
    """Between slice-resident runs the ctx keeps z / w in the kernel's own order (slice_layout.h, sl_state_index); everything
    else must keep seeing natural [B][H][W]: get_state in the middle of a run, set_state of one array only, a switch to the
    generic kernels and back, a new (smaller) problem on the same context."""
    B = 70
    masks, mid, ys = _problem(golden_inputs, B)
    args = (0.45, 0.5, 0.05, 64)
    monkeypatch.setenv('PNP_SLICE', '0')
    with P.Engine(256, 256, Bmax=B) as ref:
        ref.upload(ys, masks, mid)
        ref.init_state()
        ref.admm_cnc(3, *args)
        z3r, w3r = ref.get_state()
        ref.admm_cnc(2, *args)
        x5r = ref.x()
        z5r, w5r = ref.get_state()
    monkeypatch.delenv('PNP_SLICE')
    with P.Engine(256, 256, Bmax=B) as eng:
        eng.upload(ys, masks, mid)
        assert eng.path_name == 'slice'
        eng.init_state()
        eng.admm_cnc(3, *args)
        z3, w3 = eng.get_state()                                   # natural order out of a sliced state
        assert np.abs(z3 - z3r).max() <= 2e-5 and np.abs(w3 - w3r).max() <= 2e-5
        eng.admm_cnc(2, *args)                                     # ... and the run goes on from it
        x5 = eng.x()
        z5, w5 = eng.get_state()
        assert np.abs(x5 - x5r).max() <= 2e-5 and np.abs(z5 - z5r).max() <= 2e-5 and np.abs(w5 - w5r).max() <= 2e-5
        # one array replaced while the other stays: the untouched one must come back unchanged, in natural order
        eng.admm_cnc(1, *args)
        z6, w6 = eng.get_state()
        eng.admm_cnc(1, *args)                                     # state is sliced again
        eng.set_state(z=z3)
        zb, wb = eng.get_state()
        z7, w7 = None, None
        assert np.array_equal(zb, z3)
        eng.set_state(z=z6, w=w6)
        eng.admm_cnc(1, *args)
        z7, w7 = eng.get_state()
        assert not np.array_equal(wb, w6) and np.isfinite(wb).all()
        # the generic kernels continue from a sliced state: one generic iteration == one slice-resident iteration (round-off)
        eng.set_state(z=z6, w=w6)
        eng.admm_cnc(1, *args)                                     # slice path: state sliced
        eng.set_state(z=z6, w=w6)
        eng.set_fast_path(0)
        eng.admm_cnc(1, *args)
        zg, wg = eng.get_state()
        assert eng.path_name == 'generic' and np.abs(zg - z7).max() <= 2e-5 and np.abs(wg - w7).max() <= 2e-5
        eng.set_fast_path(1)
        eng.admm_cnc(1, *args)                                     # back on the slice path from a natural state
        assert eng.path_name == 'slice' and np.isfinite(eng.x()).all()
        # a smaller problem on the same context after a sliced state (conversion happens under the old batch size)
        eng.upload(ys[:3], masks, mid[:3])
        eng.init_state()
        eng.admm_cnc(4, *args)
        xs = eng.x()
    for b in range(3):
        assert rel_l2(xs[b], O.admm_cnc(ys[b].astype(np.complex128), masks[mid[b]], 4)) <= 2e-6


def test_a_new_problem_invalidates_the_state_of_the_one_before_it(P, golden_inputs):
    """include/pnp_mri.h: an upload makes z / w undefined until pnp_init_state or pnp_set_state(z, w).  After a slice-resident loop
    (state in the kernel's own order) a problem with ANOTHER batch size is uploaded: every entry point that would READ z / w before
    they are defined again -- get_state, the loops, set_state with only one of the two -- returns PNP_E_STATE (round 6: round 5 only
    guaranteed defined behaviour, and a run straight after the upload read the previous problem's state in the wrong order); then both
    ways of defining the state give the loop of a fresh context."""
    from pnp_admm_cnc_mri_amd import _lib
    B0, B1 = 64, 96
    masks, mid, ys = _problem(golden_inputs, B1)
    args = (0.45, 0.5, 0.05, 64)
    with P.Engine(256, 256, Bmax=B1) as fresh:
        fresh.upload(ys, masks, mid)
        fresh.init_state()
        z0, w0 = fresh.get_state()
        fresh.admm_cnc(3, *args)
        xr = fresh.x()
    with P.Engine(256, 256, Bmax=B1) as eng:
        eng.upload(ys[:B0], masks, mid[:B0])
        eng.init_state()
        eng.admm_cnc(2, *args)
        assert eng.path_name == 'slice'
        eng.upload(ys, masks, mid)                                  # B0 -> B1 with the state still in the slice order
        for call in (eng.get_state, lambda: eng.admm_cnc(1, *args), lambda: eng.admm_l1(1, 0.05, 0.5), lambda: eng.set_state(z=z0)):
            with pytest.raises(_lib.PnpError) as e:
                call()
            assert e.value.code == -3                               # PNP_E_STATE
        eng.init_state()
        z, w = eng.get_state()
        assert z.shape == (B1, 256, 256) and np.array_equal(z, z0) and not w.any()
        eng.admm_cnc(3, *args)
        assert np.array_equal(eng.x(), xr)
        eng.admm_cnc(1, *args)
        eng.upload(ys[:B0], masks, mid[:B0])                        # and B1 -> B0, state defined through set_state(z, w)
        eng.set_state(z=z0[:B0], w=w0[:B0])
        eng.admm_cnc(3, *args)
        assert np.abs(eng.x() - xr[:B0]).max() <= 2e-6


def test_driver_shaped_run_against_the_oracle_on_every_eighth_slice(P):
    """BASELINE.json configs[1] exactly as bench.py runs it for the driver (512 synthetic slices, Q_Random30, S4:176 presets,
    5 + 20 iterations as two calls, slice-resident kernel): 64 of the 512 reconstructions (every eighth slice, both rounds of
    the launch, both parities) against the float64 NumPy oracle on the same measurements.
    The bar is the north star's 1e-5.  The committed CNC presets make the map locally expansive, and a few slices amplify
    float32 round-off far more than the rest (slice 227: 2e-6 after 15 iterations, 4-6e-5 after 25 -- in NumPy's own float32
    run of the reference lines as in every kernel family here, profiles/check_slice_amplification.py): for those the bound is
    twice the NumPy-float32 control on the same measurements, and they must stay rare."""
    from pnp_admm_cnc_mri_amd import synthetic as S
    mask = S.reference_masks()['Q_Random30'].astype(np.uint8)
    B = 512
    img, noise = S.batch(0, B)
    with P.Engine(256, 256, Bmax=B) as eng:
        eng.synthesize(img, noise, mask[None], np.zeros(B, np.int32))
        assert eng.path_name == 'slice'
        y = eng.download_y()
        eng.init_state()
        eng.admm_cnc(5, 0.45, 0.5, 0.05, 64)
        eng.admm_cnc(20, 0.45, 0.5, 0.05, 64)
        x = eng.x()
    assert np.isfinite(x).all()
    beyond = []
    for b in range(3, B, 8):
        y128 = y[b].astype(np.complex128)
        ref = O.admm_cnc(y128, mask, 25)
        err = rel_l2(x[b], ref)
        if err > 1e-5:
            control = rel_l2(O.admm_cnc_f32(y128, mask, 25), ref)
            assert err <= 2 * control, (b, err, control)
            beyond.append(b)
    assert len(beyond) <= 3, beyond                    # measured: 1 of 64 (slice 227)
