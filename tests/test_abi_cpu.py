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
