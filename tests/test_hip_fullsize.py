"""The HIP path at BASELINE's full size (configs[1]: 128^3 grid, 200k particles, the scene bench.py times), checked through
properties that need no oracle run: the oracle takes ~0.1 s per substep pair at this size, the properties are exact statements
about MLS-MPM and about adjoints.  Small-size bit-level parity lives in test_hip_parity.py."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import scenarios as S  # noqa: E402

pytestmark = pytest.mark.gpu

N_GRID, N = 128, 200000


def _scene(n_grid=N_GRID, n=N, **kw):
    return S.water_block(n_grid=n_grid, n_particles=n, seed=0, **kw)


def test_largest_config_conserves_momentum(hiplib):
    """BASELINE configs' largest case (256^3 grid, 1M particles): uniform motion preserved, momentum of a stirred block conserved"""
    n = 1000000
    sc = _scene(256, n, gravity=(0.0, 0.0, 0.0))
    rng = np.random.RandomState(4)
    sc['v'] = S.f32(rng.normal(0, 0.3, (n, 3)) + [0.2, 0.1, -0.1])
    eng = S.make_engine(hiplib, sc, max_substeps_local=8)
    p0 = sc['v'].astype(np.float64).sum(0)
    st = S.run_forward(eng, 6)
    assert (st['used'] == 1).all() and np.isfinite(st['x']).all()
    assert np.abs(st['v'].astype(np.float64).sum(0) - p0).max() <= 2e-6 * np.abs(sc['v']).astype(np.float64).sum()
    assert eng.get_stats(5)['n_used'] == n
    eng.close()


def test_uniform_motion_is_preserved(hiplib):
    """P2G of a uniform velocity + G2P return it exactly (weights sum to one, C stays 0, F stays I): without gravity a block
    in uniform motion only translates (the invariants the reference's kernels satisfy by construction)."""
    sc = _scene(gravity=(0.0, 0.0, 0.0))
    v0 = S.f32([0.3, -0.2, 0.1])
    sc['v'] = np.tile(v0, (N, 1))
    eng = S.make_engine(hiplib, sc, max_substeps_local=24)
    st = S.run_forward(eng, 20)
    assert (st['used'] == 1).all()
    assert np.abs(st['v'] - v0).max() <= 2e-6
    assert np.abs(st['C']).max() <= 2e-3                      # 4/dx^2 amplifies the rounding of the weights: |C| dt stays < 1e-6
    assert np.abs(st['F'] - np.eye(3)).max() <= 1e-5
    assert np.abs(st['x'] - (sc['x'] + 20 * sc['dt'] * v0)).max() <= 2e-6
    stats = eng.get_stats(19)
    assert stats['n_used'] == N and stats['n_cells_touched'] > 30000


def test_momentum_and_particle_count_are_conserved(hiplib):
    """APIC transfers conserve linear momentum; away from the walls and without gravity the internal (pressure) forces sum to
    zero, so sum(m v) of a compressed, stirred water block does not change (equal masses: sum(v))."""
    sc = _scene(gravity=(0.0, 0.0, 0.0))
    rng = np.random.RandomState(1)
    sc['v'] = S.f32(rng.normal(0, 0.5, (N, 3)))
    sc['F'] = S.f32(np.eye(3)[None] * (1.0 + rng.uniform(-0.02, 0.02, (N, 1, 1))))         # J != 1: pressure is at work
    eng = S.make_engine(hiplib, sc, max_substeps_local=16)
    p0 = sc['v'].astype(np.float64).sum(0)
    st = S.run_forward(eng, 12)
    p1 = st['v'].astype(np.float64).sum(0)
    scale = np.abs(sc['v']).astype(np.float64).sum()
    assert np.abs(p1 - p0).max() <= 2e-6 * scale
    assert np.abs(st['v'] - sc['v']).max() > 1e-2             # the velocities did change: this is not the trivial case
    assert (st['used'] == 1).all() and np.isfinite(st['x']).all()


def test_particle_order_does_not_matter(hiplib):
    """The same particles handed over in another order give the same per-particle result (the engine sorts internally; sums
    are fp64 in LDS, so only the last bits move)."""
    sc = _scene()
    rng = np.random.RandomState(2)
    sc['v'] = S.f32(rng.normal(0, 0.3, (N, 3)))
    perm = rng.permutation(N)
    sp = dict(sc, x=sc['x'][perm], v=sc['v'][perm])
    a = S.run_forward(S.make_engine(hiplib, sc, max_substeps_local=16), 12)
    b = S.run_forward(S.make_engine(hiplib, sp, max_substeps_local=16), 12)
    assert np.abs(a['x'][perm] - b['x']).max() <= 2e-6
    assert S.rel_l2(a['v'][perm], b['v']) <= 1e-5 and S.rel_l2(a['C'][perm], b['C']) <= 1e-4


def test_adjoint_is_linear_and_matches_directional_differences(hiplib):
    """substep_grad is the transpose of the forward Jacobian: linear in the cotangent, and <grad, d> equals the directional
    derivative of <cot, state> along d (central differences in fp32 along a smooth direction)."""
    sc = _scene()
    rng = np.random.RandomState(3)
    sc['v'] = S.f32(rng.normal(0, 0.2, (N, 3)))
    K = 8
    c1, c2 = S.random_cotangent(N, seed=5), S.random_cotangent(N, seed=6)
    eng = S.make_engine(hiplib, sc, max_substeps_local=K + 2)
    _, g1 = S.run_forward_backward(eng, K, c1)
    _, g2 = S.run_forward_backward(eng, K, c2)
    mix = {k: S.f32(0.7 * c1[k] - 1.3 * c2[k]) for k in c1}
    _, gm = S.run_forward_backward(eng, K, mix)
    for k in ('gx', 'gv', 'gC', 'gF'):
        ref = 0.7 * g1[k].astype(np.float64) - 1.3 * g2[k]
        assert S.rel_l2(gm[k], ref) <= 2e-4, k
    # directional derivative along a smooth velocity perturbation d(x) = sin-field
    d = S.f32(np.stack([np.sin(7 * sc['x'][:, 1]), np.cos(5 * sc['x'][:, 2]), np.sin(3 * sc['x'][:, 0])], 1))
    eps = 2e-2

    def loss(sign):
        eng.set_frame(0, x=sc['x'], v=S.f32(sc['v'] + sign * eps * d), C_=np.zeros((N, 3, 3), np.float32), F=np.tile(np.eye(3, dtype=np.float32), (N, 1, 1)))
        st = S.run_forward(eng, K)
        return sum(float((st[a].astype(np.float64) * c1[b]).sum()) for a, b in zip('xvCF', ('gx', 'gv', 'gC', 'gF')))
    fd = (loss(+1) - loss(-1)) / (2 * eps)
    eng.set_frame(0, x=sc['x'], v=sc['v'], C_=np.zeros((N, 3, 3), np.float32), F=np.tile(np.eye(3, dtype=np.float32), (N, 1, 1)))
    _, g = S.run_forward_backward(eng, K, c1)
    an = float((g['gv'].astype(np.float64) * d).sum())
    print(f'directional derivative: central difference {fd:.6g}, adjoint {an:.6g}')
    assert abs(an) > 1.0 and abs(fd - an) <= 1e-3 * max(abs(fd), abs(an)), (fd, an)            # measured 2.7e-5
