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


def test_bench_byte_accounting_matches_the_survey():
    """bench.py credits each kernel its algorithmic bytes; they have to add up to SURVEY 8d's per-unit figures:
    forward 216 N + 72 Nc, backward 308 N + 132 Nc, pair 524 N + 204 Nc."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('bench', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    kb = bench.KERNEL_BYTES
    fwd = [kb[k] for k in bench.FWD_KERNELS]
    bwd = [kb[k] for k in ('p2g_recompute', 'grid_op_keep', 'g2p_grad', 'grid_op_grad', 'p2g_grad')]
    assert (sum(p for p, _ in fwd), sum(c for _, c in fwd)) == (216, 72)
    assert (sum(p for p, _ in bwd), sum(c for _, c in bwd)) == (308, 132)
    assert kb['sort'] == (0, 0) and kb['reorder_grad'] == (0, 0)            # layout overhead is never credited
    assert kb['g2p_p2g'] == (kb['g2p'][0] + kb['p2g'][0], kb['g2p'][1] + kb['p2g'][1])      # the fused forward launch does both kernels' work
    assert kb['pgg_g2pg'] == (kb['p2g_grad'][0] + kb['g2p_grad'][0], kb['p2g_grad'][1] + kb['g2p_grad'][1])      # ... and the fused backward launch
    assert bench.HBM_PEAK_GBS == 8000.0 and bench.N_GRID == 128 and bench.N_PARTICLES == 200000
