"""Generates tests/golden/kernel_golden.npz: outputs of the fp64 oracle (oracle/fe_oracle.cpp) on small seeded scenes.

The reference ships no vectors for this path (parity unpinned), so these are the build's own fp64 restatement,
frozen: the CPU tests check that the oracle still reproduces them (a guard against silent drift of the checker), the GPU
tests compare the HIP engine with them.  Run in the build container:   python tests/golden/make_golden_kernels.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import conftest  # noqa: E402
import scenarios as S  # noqa: E402
from fluidlab_amd._capi import EngineLib  # noqa: E402


def cases(elib, dtype=np.float64):
    """name -> dict of arrays.  Shared by the generator and the tests (which pass the library under test)."""
    out = {}
    cast = lambda c: {k: v.astype(dtype) for k, v in c.items()}
    # 1. water block, 10 substeps forward + adjoint of a random cotangent
    sc = S.water_block(n_grid=16, n_particles=800)
    sc['v'] = S.f32(np.random.RandomState(7).normal(0, 0.3, (800, 3)))
    st, g = S.run_forward_backward(S.make_engine(elib, sc), 10, cast(S.random_cotangent(sc['N'])))
    out['water'] = dict(x=st['x'], v=st['v'], C=st['C'], F=st['F'], gx=g['gx'], gv=g['gv'], gC=g['gC'], gF=g['gF'])
    # 2. every constitutive branch (liquid, viscous liquid, elastic, plasto-elastic), unused particles in between
    sc = S.mixed_materials()
    st, g = S.run_forward_backward(S.make_engine(elib, sc), 6, cast(S.random_cotangent(sc['N'])))
    out['mixed'] = dict(x=st['x'], v=st['v'], C=st['C'], F=st['F'], used=st['used'], gx=g['gx'], gv=g['gv'], gC=g['gC'], gF=g['gF'])
    # 3. the LatteArt chain in small: injector, cylinder boundary, loss, action gradient
    r = S.run_latte(elib, S.latte_mini())
    out['latte'] = dict(x=r['final']['x'], used=r['final']['used'], step_loss=r['step_loss'], action_grad=r['action_grad'], eff_state=r['eff_state'])
    # 4. a Rigid effector's moving SDF collider with a 6-dof action
    sc = S.stirrer_mini(shape='sphere', friction=0.5, softness=0.0)
    r = S.run_rigid(elib, sc, cast(S.random_cotangent(sc['N'])))
    out['rigid_effector'] = dict(x=r['final']['x'], v=r['final']['v'], action_grad=r['action_grad'], eff_state=r['eff_state'], gx0=r['gx0'])
    # 5. MAT_RIGID bodies in water
    sc = S.rigid_in_water()
    st, g = S.run_forward_backward(S.make_engine(elib, sc), 6, cast(S.random_cotangent(sc['N'])))
    out['rigid_bodies'] = dict(x=st['x'], v=st['v'], gx=g['gx'], gv=g['gv'])
    return out


if __name__ == '__main__':
    conftest._ensure_oracle()
    res = cases(EngineLib(conftest._oracle_path('f64')))
    flat = {f'{k}/{a}': np.asarray(v) for k, d in res.items() for a, v in d.items()}
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'kernel_golden.npz'), **flat)
    print({k: v.shape for k, v in flat.items()})
