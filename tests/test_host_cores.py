"""CPU tests of the __host__ __device__ cores the fused gfx950 kernels are built from
(csrc/fft16.h, csrc/fused_layout.h): compiled with g++, run lane by lane, checked against the
NumPy oracle.  Pins the two-slice packing, the mirror-column trick, the Hermitian tables and
the prox formulas before anything runs on a GPU."""
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import admm_oracle as O
from conftest import rel_l2, ROOT

SRC = os.path.join(ROOT, 'tests', 'host', 'fused_emulation.cpp')


@pytest.fixture(scope='module')
def emu(tmp_path_factory):
    d = tmp_path_factory.mktemp('emu')
    exe = str(d / 'fused_emulation')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-o', exe, SRC])
    return exe, d


def _run(emu, mode, cnc, cdc, prox, z, w, y, mask):
    exe, d = emu
    inp, out = str(d / 'in.bin'), str(d / 'out.bin')
    with open(inp, 'wb') as f:
        f.write(struct.pack('<iif5f', mode, cnc, cdc, *prox))
        for a, dt in ((z, np.float32), (w, np.float32), (y, np.complex64), (mask, np.uint8)):
            f.write(np.ascontiguousarray(a, dtype=dt).tobytes())
    subprocess.check_call([exe, inp, out])
    return np.fromfile(out, dtype=np.float32)


def _problem(golden_inputs):
    masks = np.stack([golden_inputs['masks']['Q_Random30'], golden_inputs['masks']['Q_Cartesian30']]).astype(np.uint8)
    ys = np.stack([O.synthetic_problem(b, masks[b])[1] for b in range(2)]).astype(np.complex64)
    rng = np.random.default_rng(11)
    z = rng.uniform(0, 1, (2, 256, 256)).astype(np.float32)
    w = rng.uniform(-0.1, 0.1, (2, 256, 256)).astype(np.float32)
    return z, w, ys, masks


def test_cooperative_fft256(emu, golden_inputs):
    z, w, ys, masks = _problem(golden_inputs)
    raw = _run(emu, 0, 0, 0.0, (0, 0, 0, 0, 0), z, w, ys, masks)
    fwd = raw[:2 * 65536].view(np.complex64).reshape(256, 256)
    back = raw[2 * 65536:].view(np.complex64).reshape(256, 256)
    e1 = rel_l2(fwd, np.fft.fft(ys[0].astype(np.complex128), axis=1))
    e2 = rel_l2(back / 256, ys[0])
    assert e1 <= 5e-7 and e2 <= 5e-7, (e1, e2)


@pytest.mark.parametrize('cnc', [0, 1])
def test_fused_pipeline_matches_oracle(emu, golden_inputs, cnc):
    z, w, ys, masks = _problem(golden_inputs)
    reo = 0.05
    cdc = 1.0 / (1.0 + 1.0 / 2.0 / reo)
    if cnc:
        alpha, lam, b = 0.45, 0.5, 64
        prox = (alpha * reo * lam, 1 - alpha, alpha, alpha * reo * lam * b, 1.0 / b)
    else:
        lam = 0.1
        prox = (reo * lam, 0, 0, 0, 0)
    raw = _run(emu, 1, cnc, cdc, prox, z, w, ys, masks).reshape(3, 2, 256, 256)
    for s in range(2):
        y128 = ys[s].astype(np.complex128)
        xr = O.dc_step(z[s].astype(np.float64), w[s].astype(np.float64), y128, masks[s], reo)
        if cnc:
            zr, wr = O.cnc_step(xr, z[s].astype(np.float64), w[s].astype(np.float64), alpha, lam, reo, b)
        else:
            zr, wr = O.l1_step(xr, z[s].astype(np.float64), w[s].astype(np.float64), lam, reo)
        assert rel_l2(raw[0, s], xr) <= 2e-6
        assert rel_l2(raw[1, s], zr) <= 2e-6
        assert np.abs(raw[2, s] - wr).max() <= 2e-6


def test_cooperative_fft512_structures(tmp_path):
    """32-lane 512-point transform of csrc/fft16.h: structure A (t-layout -> k-layout) and its
    transposed flow graph B, both directions, against a naive double DFT (program exits 0 when
    every relative error is < 1e-6)."""
    exe = str(tmp_path / 'fft512_emu')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-o', exe, os.path.join(ROOT, 'tests', 'host', 'fft512_emulation.cpp')])
    out = subprocess.check_output([exe]).decode()
    assert out.count('rel err') == 4


@pytest.mark.parametrize('real,tol', [('f', 5e-7), ('d', 2e-15)])
@pytest.mark.parametrize('cnc', [0, 1])
def test_split_chain_pipeline_matches_oracle(emu, golden_inputs, cnc, real, tol):
    """The "split chain" formulation of the column step (unpack the two real slices BEFORE the column
    transform, one chain per slice, repack after: csrc/fft16.h, k_fcols2) with its thread-order tables,
    in float and in double, against the oracle on float-exact inputs and hyper-parameters (measured:
    1.7e-7 in float, 3e-16 in double)."""
    exe = str(emu[1] / 'split_chain_emulation')
    if not os.path.exists(exe):
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-o', exe, os.path.join(ROOT, 'tests', 'host', 'split_chain_emulation.cpp')])
    z, w, ys, masks = _problem(golden_inputs)
    reo = 0.0625                                     # exact in float: 1/(1 + 1/(2 reo)) = 1/9 is rounded identically
    cdc = np.float32(1.0 / (1.0 + 1.0 / 2.0 / reo))
    if cnc:
        alpha, lam, b = 0.5, 0.5, 64
        prox = tuple(np.float32(v) for v in (alpha * reo * lam, 1 - alpha, alpha, alpha * reo * lam * b, 1.0 / b))
    else:
        lam = 0.125
        prox = tuple(np.float32(v) for v in (reo * lam, 0, 0, 0, 0))
    d = emu[1]
    inp, out = str(d / 'in2.bin'), str(d / 'out2.bin')
    with open(inp, 'wb') as f:
        f.write(struct.pack('<iif5f', 1, cnc, cdc, *prox))
        for a, dt in ((z, np.float32), (w, np.float32), (ys, np.complex64), (masks, np.uint8)):
            f.write(np.ascontiguousarray(a, dtype=dt).tobytes())
    subprocess.check_call([exe, inp, out, real])
    raw = np.fromfile(out, dtype=np.float64).reshape(3, 2, 256, 256)
    for s in range(2):
        y128 = ys[s].astype(np.complex128)
        z64, w64 = z[s].astype(np.float64), w[s].astype(np.float64)
        # the oracle with the float-rounded data-consistency coefficient the program was given
        v = z64 - w64
        X = np.fft.fft2(v)
        m = masks[s] != 0
        X[m] = X[m] + float(cdc) * (y128[m] - X[m])
        xr = np.abs(np.real(np.fft.ifft2(X)))
        if cnc:
            zr, wr = O.cnc_step(xr, z64, w64, alpha, lam, reo, b)
        else:
            zr, wr = O.l1_step(xr, z64, w64, lam, reo)
        assert rel_l2(raw[0, s], xr) <= tol, rel_l2(raw[0, s], xr)
        assert rel_l2(raw[1, s], zr) <= 10 * tol
        assert np.abs(raw[2, s] - wr).max() <= 10 * tol


def test_host_cores_under_address_and_ub_sanitizers(golden_inputs, tmp_path):
    """SURVEY.md section 5 (race / memory checking): the __host__ __device__ cores and the table index
    functions (the code the kernels index global memory and LDS with) built with
    -fsanitize=address,undefined and run through all three emulations: no out-of-bounds index, no
    signed overflow, no misaligned access on the CPU build (GPU sanitizers are not available on the pool)."""
    san = ['-O1', '-g', '-std=c++17', '-fsanitize=address,undefined', '-fno-sanitize-recover=all']
    host = os.path.join(ROOT, 'tests', 'host')
    exes = {}
    for name in ('fused_emulation', 'split_chain_emulation', 'slice_resident_emulation', 'fft512_emulation'):
        exes[name] = str(tmp_path / name)
        subprocess.check_call(['g++'] + san + ['-o', exes[name], os.path.join(host, name + '.cpp')])
    z, w, ys, masks = _problem(golden_inputs)
    inp = str(tmp_path / 'in.bin')
    with open(inp, 'wb') as f:
        f.write(struct.pack('<iif5f', 1, 1, 1.0 / 11.0, 0.01125, 0.55, 0.45, 0.72, 1.0 / 64))
        for a, dt in ((z, np.float32), (w, np.float32), (ys, np.complex64), (masks, np.uint8)):
            f.write(np.ascontiguousarray(a, dtype=dt).tobytes())
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    for cmd in ([exes['fused_emulation'], inp, str(tmp_path / 'o1.bin')],
                [exes['split_chain_emulation'], inp, str(tmp_path / 'o2.bin'), 'f'],
                [exes['split_chain_emulation'], inp, str(tmp_path / 'o3.bin'), 'd'],
                [exes['slice_resident_emulation'], inp, str(tmp_path / 'o4.bin')],
                [exes['fft512_emulation']]):
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0 and b'runtime error' not in r.stderr and b'AddressSanitizer' not in r.stderr, r.stderr.decode()[-1500:]
    a = np.fromfile(str(tmp_path / 'o1.bin'), dtype=np.float32).reshape(3, 2, 256, 256)[0]
    b = np.fromfile(str(tmp_path / 'o2.bin'), dtype=np.float64).reshape(3, 2, 256, 256)[0]
    assert rel_l2(a, b) <= 5e-7            # both float formulations give the same x


@pytest.mark.parametrize('cnc', [0, 1])
def test_slice_resident_pipeline_matches_oracle(emu, golden_inputs, cnc):
    """The slice-resident formulation (kernels_slice256.hip): ONE real slice as 128 packed row pairs,
    real-to-complex unpack inside the two-pass LDS transposition (csrc/slice_layout.h), 127 half-plane
    columns plus the packed column {0, 128}, Hermitian blend, and the mirrored way back -- emulated
    thread by thread with the kernel's index maps against the oracle (collisions / holes in the
    transposition maps make the program exit non-zero)."""
    exe = str(emu[1] / 'slice_resident_emulation')
    if not os.path.exists(exe):
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-o', exe, os.path.join(ROOT, 'tests', 'host', 'slice_resident_emulation.cpp')])
    z, w, ys, masks = _problem(golden_inputs)
    reo = 0.05
    cdc = 1.0 / (1.0 + 1.0 / 2.0 / reo)
    if cnc:
        alpha, lam, b = 0.45, 0.5, 64
        prox = (alpha * reo * lam, 1 - alpha, alpha, alpha * reo * lam * b, 1.0 / b)
    else:
        lam = 0.1
        prox = (reo * lam, 0, 0, 0, 0)
    d = emu[1]
    for s in range(2):                                  # the program reconstructs slice 0 of its input file
        order = [s, 1 - s]
        inp, out = str(d / 'in3.bin'), str(d / 'out3.bin')
        with open(inp, 'wb') as f:
            f.write(struct.pack('<iif5f', 1, cnc, cdc, *prox))
            for a, dt in ((z[order], np.float32), (w[order], np.float32), (ys[order], np.complex64), (masks[order], np.uint8)):
                f.write(np.ascontiguousarray(a, dtype=dt).tobytes())
        subprocess.check_call([exe, inp, out])
        raw = np.fromfile(out, dtype=np.float64).reshape(3, 256, 256)
        y128 = ys[s].astype(np.complex128)
        xr = O.dc_step(z[s].astype(np.float64), w[s].astype(np.float64), y128, masks[s], reo)
        if cnc:
            zr, wr = O.cnc_step(xr, z[s].astype(np.float64), w[s].astype(np.float64), alpha, lam, reo, b)
        else:
            zr, wr = O.l1_step(xr, z[s].astype(np.float64), w[s].astype(np.float64), lam, reo)
        assert rel_l2(raw[0], xr) <= 2e-6, rel_l2(raw[0], xr)
        assert rel_l2(raw[1], zr) <= 2e-6
        assert np.abs(raw[2] - wr).max() <= 2e-6
