"""Host logic against golden vectors produced by the reference's own NumPy code
(tests/golden/make_golden_host.py -> host_golden.npz): particle sampling, Adam, the LatteArt demo policy."""
import os
import types

import numpy as np
import pytest

from fluidlab_amd.configs.macros import COFFEE, ICECREAM, MILK, WATER
from fluidlab_amd.fluidengine.bodies import Bodies
from fluidlab_amd.optimizer.optim import Adam

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'host_golden.npz'))


def _check(tag, p):
    x = p['x']
    assert len(x) == int(G[f'{tag}_n'])
    assert (x[:64] == G[f'{tag}_head']).all() and (x[-64:] == G[f'{tag}_tail']).all()        # bit-exact
    assert np.array_equal(x.sum(0), G[f'{tag}_sum']) and np.array_equal((x * x).sum(0), G[f'{tag}_sqsum'])
    assert int(np.sum(p['used'])) == int(G[f'{tag}_used_sum']) and int(np.sum(p['mat'])) == int(G[f'{tag}_mat_sum'])
    assert float(np.sum(p['rho'])) == float(G[f'{tag}_rho_sum'])
    assert list(p['bodies']['n_particles']) == list(G[f'{tag}_body_n'])


def test_latteart_scene_particles_match_reference():
    b = Bodies(dim=3, particle_density=1e6)
    b.add_body(type='nowhere', n_particles=60000, material=MILK)
    b.add_body(type='cylinder', center=(0.5, 0.55, 0.5), height=0.1, radius=0.42, material=COFFEE)
    p = b.get()
    assert len(p['x']) == 115480                       # not ~30k
    _check('latte', p)


def test_all_samplers_match_reference():
    state = np.random.get_state()
    b = Bodies(dim=3, particle_density=2e5)
    b.add_body(type='cube', lower=(0.2, 0.2, 0.2), upper=(0.4, 0.4, 0.4), material=WATER)
    b.add_body(type='ball', center=(0.6, 0.3, 0.6), radius=0.1, material=WATER)
    b.add_body(type='cube', lower=(0.5, 0.5, 0.5), size=(0.1, 0.2, 0.1), material=ICECREAM, filling='grid', euler=(0.0, 30.0, 10.0))
    b.add_body(type='cylinder', center=(0.3, 0.7, 0.3), height=0.1, radius=0.08, material=WATER, filling='natural')
    b.add_body(type='ball', center=(0.7, 0.7, 0.3), radius=0.06, material=WATER, filling='natural')
    _check('mix', b.get())
    # add_body must leave the caller's RNG stream untouched (bodies.py:27-28,44)
    assert all(np.array_equal(a, c) for a, c in zip(state, np.random.get_state()) if isinstance(a, np.ndarray))


def test_adam_matches_reference():
    cfg = types.SimpleNamespace(lr=1e-3, beta_1=0.9, beta_2=0.99, epsilon=1e-8)
    opt = Adam((7, 3), cfg)
    rng = np.random.RandomState(int(G['adam_params0_seed']))
    params = rng.normal(size=(7, 3))
    grads = rng.normal(size=(5, 7, 3)) * np.array([1.0, 1e-3, 1e3])
    for g, ref in zip(grads, G['adam_traj']):
        params = opt.step(params, g)
        assert np.array_equal(params, ref)             # same fp64 arithmetic, bit for bit


def test_latteart_demo_policy_matches_reference():
    from fluidlab_amd.envs.latteart_env import LatteArtEnv
    fake = types.SimpleNamespace(horizon_action=250, agent=types.SimpleNamespace(action_dim=3))
    pol = LatteArtEnv.demo_policy(fake)
    assert np.abs(pol.actions_v - G['latte_demo_actions_v']).max() < 1e-15
    assert np.array_equal(pol.actions_p, G['latte_demo_actions_p'])


def test_pose_conventions_match_reference():
    """utils/geom.py's NumPy helpers, Effector.init_rot (effector.py:45) and Mesh.init_transform (mesh.py:97-103) against the
    reference's own outputs: euler (degrees, 'zyx' of the reversed triple) -> wxyz quaternion -> T_init = trans * rot * scale."""
    from fluidlab_amd.utils import geom
    from fluidlab_amd.fluidengine.meshes import Static, sdf_sphere
    from fluidlab_amd.fluidengine.effectors import Rigid
    from fluidlab_amd.configs.macros import PLATE
    for row in G['pose_cases']:
        pos, euler, scale, quat = row[0:3], row[3:6], row[6:9], row[9:13]
        T_init = row[13:29].reshape(4, 4)
        moved, pts, rotated = row[29:44].reshape(5, 3), row[44:59].reshape(5, 3), row[59:74].reshape(5, 3)
        q = geom.euler_to_quat_wxyz(euler)
        assert np.abs(q - quat).max() < 1e-14 or np.abs(q + quat).max() < 1e-14
        assert np.abs(geom.transform_by_quat_np(pts, quat) - rotated).max() < 1e-13
        # a collider built from the same pose maps world points back into the mesh frame with inverse(T_init)
        st = Static(material=PLATE, file=None, sdf=sdf_sphere(0.3), sdf_res=8, pos=tuple(pos), euler=tuple(euler), scale=tuple(scale), has_dynamics=True)
        from fluidlab_amd.utils.mesh import sdf_lattice
        T_lat = sdf_lattice(8)[1]
        assert np.abs(st.T_mesh_to_voxels_np - T_lat @ np.linalg.inv(T_init)).max() < 1e-9
        back = (moved @ np.linalg.inv(T_init)[:3, :3].T) + np.linalg.inv(T_init)[:3, 3]
        assert np.abs(back - pts).max() < 1e-12
        # an effector initialised with the same euler angles starts in that orientation
        eff = Rigid(max_substeps_local=4, max_substeps_global=4, max_action_steps_global=2, ckpt_dest='cpu', init_pos=tuple(pos), init_euler=tuple(euler),
                    action_dim=3, action_scale_p=(1, 1, 1), action_scale_v=(1, 1, 1))
        r = np.asarray(eff.init_rot, np.float64)
        assert np.abs(r - quat).max() < 1e-6 or np.abs(r + quat).max() < 1e-6


def test_normalize_mesh_matches_reference():
    from fluidlab_amd.utils.mesh import TriMesh, normalize_mesh
    a, b = TriMesh(G['normalize_in_a'], [[0, 1, 2]]), TriMesh(G['normalize_in_b'], [[0, 1, 2]])
    assert np.array_equal(normalize_mesh(a).vertices, G['normalize_self'])
    assert np.array_equal(normalize_mesh(a, b).vertices, G['normalize_by_other'])


def test_material_tables_match_reference():
    from fluidlab_amd.configs import macros as M
    for name, mid, row in zip(G['mat_names'], G['mat_ids'], G['mat_table']):
        assert getattr(M, str(name)) == int(mid)
        assert [M.MU[int(mid)], M.LAMDA[int(mid)], M.RHO[int(mid)], M.MAT_CLASS[int(mid)]] == list(row), name
    for name, oid, fr in zip(G['friction_names'], G['friction_ids'], G['friction']):
        assert getattr(M, str(name)) == int(oid) and M.FRICTION[int(oid)] == float(fr), name
    assert list(M.NOWHERE) == list(G['nowhere']) and M.EPS == float(G['eps'])


def test_staged_policies_match_reference():
    from fluidlab_amd.optimizer import policies as P
    optim_cfg = types.SimpleNamespace(type='Adam', lr=1e-3, beta_1=0.9, beta_2=0.99, epsilon=1e-8)
    for cls, dim, H in ((P.GatheringPolicy, 3, 300), (P.GatheringOPolicy, 3, 300), (P.MixingPolicy, 3, 200), (P.TransportingPolicy, 6, 50),
                        (P.IceCreamDynamicPolicy, 3, 900), (P.IceCreamStaticPolicy, 3, 50)):
        ir = types.SimpleNamespace(v=(np.zeros(dim), np.zeros(dim)), p=(np.full(dim, 0.5), np.full(dim, 0.5)))
        pol = cls(optim_cfg, ir, dim, H, np.array([-0.01, 0.01]), fix_dim=[1])
        assert np.array_equal(np.asarray(pol.trainable), G[f'policy_{cls.__name__}_trainable']), cls.__name__
        if f'policy_{cls.__name__}_status' in G:
            assert np.array_equal(np.asarray(pol.status), G[f'policy_{cls.__name__}_status']), cls.__name__
    pol = P.TransportingPolicy(optim_cfg, types.SimpleNamespace(v=(np.zeros(6), np.zeros(6)), p=(np.full(6, 0.5), np.full(6, 0.5))), 6, 8,
                               np.array([-0.0005, 0.0005]), fix_dim=[1, 2, 3, 4])
    pol.optimize(G['policy_step_grads'], {'temporal_range': 8})
    assert np.array_equal(np.vstack([pol.actions_v, pol.actions_p[None]]), G['policy_step_result'])


def test_later_env_bodies_match_reference():
    from fluidlab_amd.configs.macros import MILK_VIS, RIGID_HEAVY
    b = Bodies(dim=3, particle_density=1e6)
    b.add_body(type='nowhere', n_particles=1000, material=WATER)
    b.add_body(type='cube', lower=(0.275, 0.475, 0.475), size=(0.05, 0.05, 0.05), euler=(45.0, 45.0, 45.0), color=(1.0, 0.5, 0.5, 1.0), filling='natural',
               material=RIGID_HEAVY)
    b.add_body(type='cube', lower=(0.425, 0.55, 0.425), upper=(0.575, 0.7, 0.575), material=MILK_VIS)
    _check('later', b.get())
