"""Host build of fluidlab_amd/csrc/fe_math.h -- the exact math the HIP kernels inline -- checked on the CPU:
fp64: finite-difference check of the hand-derived constitutive/SVD adjoints (mpm:272-292, 326-401);
fp32: the SVD contract (U S V^T = F, orthogonality, descending sigma, det signs) at the product precision.
No GPU and no oracle involved: hipcc --offload-host-only compiles the header's __host__ __device__ functions for x86."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tests', 'csrc', 'math_test.cpp')
OUT = os.path.join(ROOT, 'tests', 'csrc', '_build')


def _hipcc():
    return '/opt/rocm/bin/hipcc' if os.path.exists('/opt/rocm/bin/hipcc') else shutil.which('hipcc')


@pytest.mark.parametrize('real', ['double', 'float'])
def test_fe_math_host(real):
    hipcc = _hipcc()
    if hipcc is None:
        pytest.skip('hipcc not available')
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join(OUT, f'math_test_{real}')
    subprocess.check_call([hipcc, '--offload-host-only', '-O2', '-std=c++17', f'-DFE_T={real}', '-x', 'hip', SRC, '-o', exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert '0 failures' in r.stdout
