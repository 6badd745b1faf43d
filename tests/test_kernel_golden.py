"""The committed fp64 outputs of the oracle (tests/golden/kernel_golden.npz, made by tests/golden/make_golden_kernels.py) as a
fixed reference point: the oracle must keep reproducing them (CPU), and the HIP engine is compared with them (GPU) under the
tolerances of the live parity tests.  They are this build's own restatement frozen -- the reference
ships no vectors for this path."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
import scenarios as S  # noqa: E402
from make_golden_kernels import cases  # noqa: E402

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'kernel_golden.npz'))


def _gold(case):
    return {k.split('/', 1)[1]: G[k] for k in G.files if k.startswith(case + '/')}


def test_oracle_f64_reproduces_the_fixture(oracle64):
    got = cases(oracle64)
    assert sorted(f'{c}/{a}' for c, d in got.items() for a in d) == sorted(G.files)
    for c, d in got.items():
        for a, v in d.items():
            ref = G[f'{c}/{a}']
            if ref.dtype.kind in 'iu':
                assert (np.asarray(v) == ref).all(), (c, a)
            else:                                         # OpenMP sums: the order of additions may differ, nothing else
                assert np.abs(np.asarray(v, np.float64) - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), (c, a)


def _check(got, tol_x, tol_rel, grad_cos, grad_rel):
    for c, d in got.items():
        ref = _gold(c)
        if 'used' in ref:
            assert (np.asarray(d['used']) == ref['used']).all(), c
        assert np.abs(d['x'] - ref['x']).max() <= tol_x, (c, np.abs(d['x'] - ref['x']).max())
        for a in ('v', 'C', 'F', 'step_loss'):
            if a in ref:
                err = np.abs(np.asarray(d[a], np.float64) - ref[a]).max()
                assert S.rel_l2(d[a], ref[a]) <= tol_rel and err <= 10 * tol_rel * max(1.0, np.abs(ref[a]).max()), (c, a, S.rel_l2(d[a], ref[a]), err)
        if 'eff_state' in ref:
            assert np.abs(d['eff_state'] - ref['eff_state']).max() <= 1e-6, c
        for a in ('gx', 'gv', 'gC', 'gF', 'gx0', 'action_grad'):
            if a in ref:
                assert S.cosine(d[a], ref[a]) >= grad_cos and S.rel_l2(d[a], ref[a]) <= grad_rel, (c, a, S.cosine(d[a], ref[a]), S.rel_l2(d[a], ref[a]))


def test_oracle_f32_tracks_the_fixture(oracle32):
    _check(cases(oracle32, np.float32), tol_x=1e-5, tol_rel=5e-3, grad_cos=0.999, grad_rel=2e-2)


@pytest.mark.gpu
def test_hip_matches_the_fixture(hiplib):
    """fp32 HIP engine against the frozen fp64 outputs: positions to 1e-5, state to 5e-3 relative, adjoints cos >= 0.999"""
    _check(cases(hiplib, np.float32), tol_x=1e-5, tol_rel=5e-3, grad_cos=0.999, grad_rel=2e-2)
