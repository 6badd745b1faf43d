import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _oracle_path(prec):
    return os.path.join(ROOT, 'oracle', '_build', f'libfe_oracle_{prec}.so')


def _ensure_oracle():
    if not (os.path.exists(_oracle_path('f32')) and os.path.exists(_oracle_path('f64'))):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')])


@pytest.fixture(scope='session')
def oracle64():
    """fp64 build of the CPU restatement (the parity oracle)."""
    from fluidlab_amd._capi import EngineLib
    _ensure_oracle()
    return EngineLib(_oracle_path('f64'))


@pytest.fixture(scope='session')
def oracle32():
    from fluidlab_amd._capi import EngineLib
    _ensure_oracle()
    return EngineLib(_oracle_path('f32'))


@pytest.fixture(scope='session')
def hiplib():
    """The product library; only usable on a GPU box."""
    from fluidlab_amd import _capi
    alt = os.environ.get('FE_TEST_HIP_LIB')          # A/B builds of the engine (scripts/ab_bench.py): tests only, never the product path
    if alt:
        _capi.HIP_LIB_PATH = os.path.abspath(alt)
    return _capi.load_hip()
