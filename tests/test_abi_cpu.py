"""The C-ABI library loads on a machine without a GPU and exports exactly what include/pnp_mri.h
declares; argument errors come back as codes + messages, never as crashes (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from pnp_admm_cnc_mri_amd import _lib

HEADER = os.path.join(ROOT, 'include', 'pnp_mri.h')


def _declared():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(pnp_[A-Za-z0-9_]+)\s*\(', src)))


def test_header_and_binding_agree():
    names = _declared()
    assert len(names) >= 30
    assert sorted(_lib.SIGNATURES) == names


def test_every_declared_symbol_is_exported():
    L = _lib.lib()
    for n in _declared():
        assert hasattr(L, n), n
    assert L.pnp_abi_version() == _lib.ABI_VERSION == int(re.search(r'#define PNP_ABI_VERSION\s+(\d+)', open(HEADER).read()).group(1))


def test_argument_errors_are_codes_not_crashes():
    L = _lib.lib()
    ctx = _lib.ctx_p()
    assert L.pnp_ctx_create(0, 100, 256, 1, C.byref(ctx)) == -1            # PNP_E_ARG
    assert b'256 or 512' in L.pnp_last_error()
    assert L.pnp_ctx_create(0, 256, 256, 0, C.byref(ctx)) == -1
    assert L.pnp_ctx_create(0, 256, 256, 1, None) == -1
    assert L.pnp_init_state(None) == -1 and L.pnp_admm_l1_run(None, 1, 0.1, 0.1) == -1
    assert L.pnp_ctx_destroy(None) == 0
    n = C.c_int(-1)
    assert L.pnp_device_count(C.byref(n)) == 0 and n.value >= 0
    with pytest.raises(_lib.PnpError):
        _lib.check(L.pnp_sync(None))


def test_environment_knobs_are_validated_at_context_creation(monkeypatch):
    """A mistyped or out-of-range knob fails pnp_ctx_create with PNP_E_ARG and names the variable (checked before any
    HIP call, so this runs without a GPU); the experiment knobs of the profiling scripts are compiled out of the
    product library and therefore ignored."""
    L = _lib.lib()
    ctx = _lib.ctx_p()
    for name, bad in (('PNP_SLICE', '2'), ('PNP_SLICE', 'yes'), ('PNP_SLICE_MIN_B', '0'), ('PNP_SLICE_PAD_KB', '-1'),
                      ('PNP_SLICE_YH_PAD_KB', '4k'), ('PNP_FUSED_STREAMS', '9'), ('PNP_FUSED_COLS', '3'), ('PNP_FUSED_CHUNK', '')):
        monkeypatch.setenv(name, bad)
        assert L.pnp_ctx_create(0, 256, 256, 1, C.byref(ctx)) == -1, (name, bad)
        assert name.encode() in L.pnp_last_error()
        monkeypatch.delenv(name)
    # experiment knob with a value that an experiment build would reject: the product library never reads it
    monkeypatch.setenv('PNP_SLICE_QUEUES', 'many')
    rc = L.pnp_ctx_create(0, 256, 256, 1, C.byref(ctx))
    assert rc in (0, -2)                                     # created on a GPU box, HIP error without a device -- never PNP_E_ARG
    if rc == 0:
        L.pnp_ctx_destroy(ctx)
    import subprocess
    syms = subprocess.check_output(['strings', _lib.LIB_PATH]).decode()
    for knob in ('PNP_SLICE_XOR', 'PNP_SLICE_SEGMENT', 'PNP_SLICE_QUEUES', 'PNP_SLICE_FLIP', 'PNP_F512_QUEUES', 'PNP_F256S_QUEUES',
                 'PNP_GENERIC_STOCKHAM', 'PNP_SLICE_PROF'):
        assert knob not in syms, knob


def test_no_cpu_fallback_when_library_is_missing(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'libpnpmri.so'))
    with pytest.raises(ImportError, match='no CPU fallback'):
        _lib.lib()


def test_engine_without_gpu_fails_loudly():
    """On a box with no HIP device the product path raises; it never computes on the CPU."""
    if _lib.device_count() > 0:
        pytest.skip('a GPU is present')
    import pnp_admm_cnc_mri_amd as P
    with pytest.raises(_lib.PnpError):
        P.Engine(256, 256, 1)


def _build_consumer(tmp_path):
    import subprocess
    exe = str(tmp_path / 'abi_consumer')
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-I', os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'tests', 'host', 'abi_consumer.c'), '-o', exe,
                           '-L', os.path.dirname(_lib.LIB_PATH), '-lpnpmri', '-lm',
                           '-Wl,-rpath,' + os.path.dirname(_lib.LIB_PATH)])
    return exe


def test_plain_c_consumer_links_and_gets_error_codes(tmp_path):
    """include/pnp_mri.h compiles as C99 and a gcc-built program drives the library."""
    import subprocess
    exe = _build_consumer(tmp_path)
    out = subprocess.check_output([exe]).decode()
    assert out.startswith('devices ')
