"""The C-ABI libraries load and export every symbol include/fluidengine.h declares.
No compute is called here (no GPU in the build container)."""
import os
import re

import pytest

from fluidlab_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, 'include', 'fluidengine.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(fe_[a-z_0-9]+)\s*\(', src)))


def test_binding_covers_header():
    assert _header_symbols() == sorted(_capi.ABI_SYMBOLS)


def test_hip_library_exports_abi():
    if not os.path.exists(_capi.HIP_LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _capi.load_hip()
    assert lib.backend == 'hip-gfx950' and lib.real_size == 4
    assert lib.missing_symbols() == []


def test_oracle_libraries_export_abi(oracle32, oracle64):
    assert oracle32.backend == 'oracle-f32' and oracle32.missing_symbols() == []
    assert oracle64.backend == 'oracle-f64' and oracle64.missing_symbols() == []


def test_hip_engine_fails_loudly_without_gpu():
    """No silent CPU fallback: without a HIP device fe_create must fail with a clear message."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is visible here')
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    import scenarios as S
    lib = _capi.load_hip()
    with pytest.raises(_capi.FeEngineError, match='no HIP device|no CPU fallback'):
        S.make_engine(lib, S.water_block(n_grid=8, n_particles=8))
