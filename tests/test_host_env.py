"""The Python host layer (MPMSimulator / TaichiEnv / Agent / Loss / Solver / envs) on CPU.

The product path loads the HIP library and has no fallback; these tests hand the same host code the oracle
build of the ABI (`engine_lib=`) so time indexing, action buffering, the chunked checkpoint protocol, the loss
curriculum and the optimiser loop can be exercised without a GPU."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))

from fluidlab_amd.envs import make
from fluidlab_amd.optimizer.recorder import Recorder
from fluidlab_amd.optimizer.solver import Solver
from fluidlab_amd.utils.config import load_config

MINI = dict(quality=0.5, particle_density=4e4, n_pool=300, horizon=12, horizon_action=8)


def _cfg(n_iters=2):
    cfg = load_config('configs/exp_latteart.yaml')
    assert cfg.SOLVER.optim.lr == 1e-3 and cfg.SOLVER.init_range.p[0] == (0.15, 0.65, 0.5)
    cfg.SOLVER.n_iters = n_iters
    return cfg.SOLVER


@pytest.fixture(scope='module')
def mini_target(oracle32):
    env = make('LatteArt-v0', seed=0, loss=False, engine_lib=oracle32, **MINI)
    return Recorder(env).record(write=False)


def test_record_then_optimise(oracle32, mini_target):
    tgt = mini_target
    assert len(tgt['x']) == MINI['horizon'] and tgt['x'][0].shape == (2503, 3)
    # 8 action steps x 10 substeps x flux 2 particles were injected
    assert int(tgt['used'][-1].sum()) - int((tgt['mat'] == 2).sum()) == 8 * 10 * 2
    env = make('LatteArt-v0', seed=0, loss=True, target=tgt, engine_lib=oracle32, **MINI)
    losses = []
    Solver(env, None, _cfg(3)).solve(callback=lambda it, info, pol: losses.append(info['loss']))
    assert losses[0] > losses[1] > losses[2] > 0


def _batch_vs_single(lib, mini_target, seeds=(3, 4)):
    """EnvBatch.forward_backward over two replicas (own injector noise) against Solver.forward_backward on each of them alone."""
    from fluidlab_amd.optimizer.batch import EnvBatch

    def build(seed):
        env = make('LatteArt-v0', seed=seed, loss=True, target=mini_target, engine_lib=lib, **MINI)
        np.random.seed(0)                                         # the same initial policy for every replica
        pol = env.trainable_policy(_cfg().optim, _cfg().init_range)
        pol.actions_v[:] = np.random.RandomState(4).uniform(-0.004, 0.004, pol.actions_v.shape)
        return env, pol

    single = []
    for sd in seeds:
        env, pol = build(sd)
        single.append(Solver(env, None, _cfg()).forward_backward(env.taichi_env.get_state()['state'], pol, env.horizon, env.horizon_action))
    envs, pols = zip(*[build(sd) for sd in seeds])
    batch = EnvBatch(envs).forward_backward([e.taichi_env.get_state()['state'] for e in envs], pols, envs[0].horizon, envs[0].horizon_action)
    return single, batch


def test_env_batch_equals_environments_stepped_alone(oracle64, mini_target):
    """optimizer/batch.py: B environments through ONE engine call per step (fe_step_batch) give each environment exactly what it gets
    alone -- loss, action gradient -- and the replicas differ (their injector noise does)."""
    single, batch = _batch_vs_single(oracle64, mini_target)
    for (ia, ga), (ib, gb) in zip(single, batch):
        assert ia['loss'] == ib['loss'] and np.array_equal(ga, gb)
    assert single[0][0]['loss'] != single[1][0]['loss'] and np.abs(batch[0][1]).max() > 0


def test_chunked_checkpointing_equals_resident_trajectory(oracle64, mini_target, tmp_path, monkeypatch):
    """mpm:777-912: backward through 20-substep chunks (checkpoint + re-forward) must give the gradient of the
    whole-trajectory-resident mode."""
    from fluidlab_amd.fluidengine.effectors import Injector
    base = np.random.RandomState(7).uniform(size=(20, 2, 3))
    # `locally_random` noise is indexed by the local frame: make it 20-periodic so every chunking sees the same noise
    monkeypatch.setattr(Injector, 'random_vector_factory', staticmethod(lambda n, flux, dim: np.tile(base, (n // 20 + 1, 1, 1))[:n]))
    grads = {}
    for mode, kw in [('resident', dict(max_substeps_local=None)), ('cpu', dict(max_substeps_local=20, ckpt_dest='cpu')),
                     ('disk', dict(max_substeps_local=40, ckpt_dest='disk'))]:
        env = make('LatteArt-v0', seed=0, loss=True, target=mini_target, engine_lib=oracle64, **MINI, **kw)
        solver = Solver(env, None, _cfg())
        policy = env.trainable_policy(_cfg().optim, _cfg().init_range)
        policy.actions_v[:] = np.random.RandomState(4).uniform(-0.004, 0.004, policy.actions_v.shape)
        info, g = solver.forward_backward(env.taichi_env.get_state()['state'], policy, env.horizon, env.horizon_action)
        grads[mode] = (info['loss'], g)
        if mode == 'disk':
            # the chunk files follow the reference's wire format (mpm:777-803, effector.py:103-110, injector.py:131-140):
            # <start_substep:06d>.pkl holding x, v, C, F, used, actions and one {pos, quat, v, w, act_id} per effector
            import os
            import pickle
            sim = env.taichi_env.simulator
            files = sorted(os.listdir(sim.ckpt_dir))
            assert files and all(len(f) == 10 and f.endswith('.pkl') and f[:6].isdigit() for f in files)
            assert [int(f[:6]) for f in files] == list(range(0, 40 * len(files), 40))
            ck = pickle.load(open(os.path.join(sim.ckpt_dir, files[0]), 'rb'))
            assert set(ck) == {'x', 'v', 'C', 'F', 'used', 'actions', 'agent'}
            N = sim.n_particles
            assert ck['x'].shape == (N, 3) and ck['C'].shape == (N, 3, 3) and ck['used'].dtype == np.int32
            assert len(ck['actions']) == 40 // sim.n_substeps
            assert set(ck['agent'][0]) == {'pos', 'quat', 'v', 'w', 'act_id'}
            assert ck['agent'][0]['pos'].shape == (3,) and ck['agent'][0]['quat'].shape == (4,)
    for mode in ('cpu', 'disk'):
        assert abs(grads[mode][0] - grads['resident'][0]) < 1e-9 * abs(grads['resident'][0])
        assert np.abs(grads[mode][1] - grads['resident'][1]).max() < 1e-9 * np.abs(grads['resident'][1]).max()


def test_rl_api_and_state_roundtrip(oracle32, mini_target):
    env = make('LatteArt-v0', seed=0, loss=True, target=mini_target, engine_lib=oracle32, **MINI)
    obs0 = env.reset()
    assert obs0.shape == env.observation_space.shape and env.action_space.shape == (3,)
    obs, reward, done, _ = env.step(np.array([0.5, 0.0, -0.5]))          # clipped to +-0.05
    assert np.isfinite(reward) and not done and env.t == 1
    sim = env.taichi_env.simulator
    assert sim.cur_substep_global == 10 and sim.cur_step_global == 1 and sim.cur_substep_local == 10
    st = env.taichi_env.get_state()
    assert st['state']['agent'][0].shape == (8,) and st['state']['agent'][0][7] == 0 + 20      # act_id = act_range[0] (the first pool id, here 0) + 10 substeps x flux 2
    obs1 = env.reset()
    assert np.array_equal(obs0, obs1)


def test_time_indexing_asserts():
    from fluidlab_amd.fluidengine.simulators import MPMSimulator
    sim = MPMSimulator(dim=3, quality=1, gravity=(0, -10, 0), horizon=330, max_substeps_local=50, max_substeps_global=100000, ckpt_dest='cpu')
    assert (sim.n_grid, sim.n_substeps, sim.max_steps_local) == (64, 10, 5) and sim.p_vol == (0.5 / 64) ** 2
    sim.cur_substep_global = 3217
    assert (sim.cur_substep_local, sim.cur_step_local, sim.cur_step_global) == (17, 1, 321)
    with pytest.raises(AssertionError):
        MPMSimulator(dim=3, quality=1, gravity=(0, -10, 0), horizon=10, max_substeps_local=45, max_substeps_global=1000, ckpt_dest='cpu')
    resident = MPMSimulator(dim=3, quality=2, gravity=(0, -10, 0), horizon=330, max_substeps_local=None, max_substeps_global=100000, ckpt_dest='cpu')
    assert resident.max_substeps_local == 3310 and resident.n_grid == 128


def test_product_path_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is visible here')
    from fluidlab_amd._capi import FeEngineError
    with pytest.raises(FeEngineError, match='no HIP device|no CPU fallback'):
        make('WaterBlock-v0', quality=0.5, n_particles=500, horizon=2)


def test_waterblock_env_matches_bench_scene(oracle32):
    import scenarios as S
    env = make('WaterBlock-v0', quality=0.5, n_particles=4096, horizon=3, engine_lib=oracle32)
    sc = S.water_block(n_grid=32, n_particles=4096)
    assert np.array_equal(env.taichi_env.simulator.get_x(0), sc['x'])      # same RNG stream as bench.py / SURVEY 8d C2


def test_rigid_body_through_the_python_stack(oracle64):
    """A MAT_RIGID body added with TaichiEnv.add_body reaches the engine with its body id (mpm:176-201) and stays
    rigid while it falls into water (shape matching, mpm:449-505)."""
    from fluidlab_amd.configs.macros import RIGID, WATER
    from fluidlab_amd.fluidengine.taichi_env import TaichiEnv
    np.random.seed(0)
    te = TaichiEnv(dim=3, quality=0.25, particle_density=2e4, horizon=4, gravity=(0.0, -10.0, 0.0), engine_lib=oracle64)
    te.setup_boundary(type='cube', lower=(0.05, 0.05, 0.05), upper=(0.95, 0.95, 0.95))
    te.add_body(type='cube', lower=(0.3, 0.2, 0.3), upper=(0.7, 0.4, 0.7), material=WATER)
    te.add_body(type='cube', lower=(0.4, 0.42, 0.4), upper=(0.6, 0.55, 0.6), material=RIGID)
    te.build()
    sim = te.simulator
    assert sim.n_bodies == 2
    x0 = sim.get_x(0).astype(np.float64)
    rigid = np.asarray(te.particles['body_id']) == 1
    assert rigid.sum() > 20
    for _ in range(3):
        te.step(None)
    x1 = te.get_state()['state']['x'].astype(np.float64)
    d0 = np.linalg.norm(x0[rigid][:, None] - x0[rigid][None], axis=2)
    d1 = np.linalg.norm(x1[rigid][:, None] - x1[rigid][None], axis=2)
    assert np.abs(d1 - d0).max() < 1e-10
    assert (x0[rigid][:, 1] - x1[rigid][:, 1]).min() > 1e-4        # it fell


def _cup_scene(engine_lib, quality=0.5, density=1.5e5):
    from fluidlab_amd.configs.macros import CUP, WATER
    from fluidlab_amd.fluidengine.meshes import sdf_cup
    from fluidlab_amd.fluidengine.taichi_env import TaichiEnv
    np.random.seed(0)
    te = TaichiEnv(dim=3, quality=quality, particle_density=density, horizon=140, gravity=(0.0, -10.0, 0.0), engine_lib=engine_lib)
    te.setup_boundary(type='cube', lower=(0.05, 0.05, 0.05), upper=(0.95, 0.95, 0.95))
    # a cup of outer radius 0.25 and height 0.3 standing at y in [0.2, 0.5], scaled/placed like a reference static
    te.add_static(file='cup.obj', material=CUP, has_dynamics=True, sdf=sdf_cup(0.5, 0.5, 0.2), sdf_res=48,        # walls >= 2 dx thick: contact acts on grid nodes
                  pos=(0.5, 0.35, 0.5), euler=(0.0, 0.0, 0.0), scale=(0.5, 0.3, 0.5))
    te.add_body(type='cube', lower=(0.44, 0.40, 0.44), upper=(0.56, 0.62, 0.56), material=WATER)
    te.build()
    return te


def test_static_cup_holds_water(oracle32):
    """TaichiEnv.add_static -> Static (mesh.py:97-127 pose handling) -> engine colliders: water dropped into an analytic
    cup stays inside it instead of falling to the floor of the domain."""
    te = _cup_scene(oracle32)
    st = te.statics[0]
    assert st.sdf([[0.5, 0.22, 0.5]])[0] < 0 < st.sdf([[0.5, 0.40, 0.5]])[0]        # bottom plate solid, cavity empty
    assert st.sdf([[0.5 + 0.2, 0.35, 0.5]])[0] < 0                                  # side wall solid
    for _ in range(130):
        te.step(None)
    x = te.get_state()['state']['x']
    r = np.hypot(x[:, 0] - 0.5, x[:, 2] - 0.5)
    assert np.isfinite(x).all()
    assert x[:, 1].min() > 0.2                      # nothing fell through the bottom (floor of the cup at y = 0.2 + wall)
    assert (r < 0.25).mean() > 0.97                 # and the walls hold it
    # without the cup the same water reaches the domain floor
    from fluidlab_amd.fluidengine.taichi_env import TaichiEnv
    from fluidlab_amd.configs.macros import WATER
    np.random.seed(0)
    free = TaichiEnv(dim=3, quality=0.5, particle_density=1.5e5, horizon=140, gravity=(0.0, -10.0, 0.0), engine_lib=oracle32)
    free.setup_boundary(type='cube', lower=(0.05, 0.05, 0.05), upper=(0.95, 0.95, 0.95))
    free.add_body(type='cube', lower=(0.44, 0.40, 0.44), upper=(0.56, 0.62, 0.56), material=WATER)
    free.build()
    for _ in range(130):
        free.step(None)
    assert free.get_state()['state']['x'][:, 1].min() < 0.2


def _stir_env(engine_lib, horizon=6):
    """A LatteArtStir-like scene: water in the domain, an AgentRigid whose stirrer is an analytic cylinder SDF."""
    from fluidlab_amd.configs.macros import STIRRER, WATER
    from fluidlab_amd.fluidengine.meshes import sdf_cylinder
    from fluidlab_amd.fluidengine.taichi_env import TaichiEnv
    from fluidlab_amd.utils.config import CfgNode
    np.random.seed(0)
    te = TaichiEnv(dim=3, quality=0.5, particle_density=6e4, horizon=horizon, gravity=(0.0, -10.0, 0.0), engine_lib=engine_lib,
                   max_substeps_local=None)
    te.setup_boundary(type='cube', lower=(0.05, 0.05, 0.05), upper=(0.95, 0.95, 0.95))
    te.setup_agent(CfgNode(dict(type='AgentRigid', effectors=[dict(
        type='Rigid', params=dict(init_pos=(0.5, 0.32, 0.5), init_euler=(0.0, 0.0, 0.0), action_dim=6,
                                  action_scale_p=(1.0,) * 6, action_scale_v=(1.0,) * 6),
        mesh=dict(file='stirrer.obj', material=STIRRER, softness=0.0, sdf=sdf_cylinder(0.25, 0.5), sdf_res=40,
                  scale=(0.3, 0.4, 0.3), euler=(0.0, 0.0, 15.0)),
        boundary=dict(type='cube', lower=(0.05, 0.05, 0.05), upper=(0.95, 0.95, 0.95)))])))
    te.add_body(type='cube', lower=(0.3, 0.1, 0.3), upper=(0.7, 0.3, 0.7), material=WATER)
    te.build()
    return te


def test_agent_rigid_stirs_water(oracle64):
    """Rigid + Dynamic mesh + AgentRigid through TaichiEnv: the stirrer drags water along (STIRRER friction 8 < 10 takes
    the Coulomb branch) and a 6-dof action gradient comes back through step_grad."""
    te = _stir_env(oracle64)
    sim = te.simulator
    x0 = sim.get_x(0).copy()
    te.set_state(te.get_state()['state'], grad_enabled=True)
    actions = np.tile([0.02, 0.0, 0.0, 0.0, 0.3, 0.0], (6, 1))
    for a in actions:
        te.step(a)
    x1 = te.get_state()['state']['x']
    near = np.hypot(x0[:, 0] - 0.5, x0[:, 2] - 0.5) < 0.1
    assert (x1[near, 0] - x0[near, 0]).mean() > 3 * abs((x1[~near, 0] - x0[~near, 0]).mean())    # dragged along +x
    # gradient of sum(x) at the end w.r.t. the actions
    te.reset_grad()
    sim.engine.add_grad(sim.cur_substep_local, np.ones_like(x1), None, None, None)
    for a in actions[::-1]:
        te.step_grad(a)
    g = te.agent.get_grad(6)
    assert g.shape == (7, 6) and np.isfinite(g).all()
    assert np.abs(g[:6, 0]).min() > 0 and np.abs(g[:6, 4]).max() > 0          # translation and rotation both matter


def _circulation(engine_lib, **kw):
    det = [[6, 16, 21], [9, 16, 21], [6, 16, 10], [26, 16, 16], [24, 16, 20], [20, 16, 8]]
    return make('Circulation-v0', seed=0, loss=True, res=32, horizon=6, solver_iters=10, detectors=det, engine_lib=engine_lib, **kw)


def test_circulation_env_smoke_field(oracle64):
    """Circulation-v0 through the whole stack: SmokeField + AirCon + CirculationLoss + room SDF.  The AirCon cools the air
    it blows at, the loss gradient w.r.t. its 8-dof actions is non-trivial, and backward through 2-step checkpoint chunks
    (smoke frames copied / reloaded with the MPM state, mpm:794-866) equals the resident trajectory."""
    grads = {}
    for mode, kw in [('resident', dict(max_substeps_local=None)), ('chunked', dict(max_substeps_local=20, ckpt_dest='cpu'))]:
        env = _circulation(oracle64, **kw)
        te = env.taichi_env
        sf = te.smoke_field
        assert (sf.lower_y, sf.higher_y) == (15, 17)
        cfg = load_config('configs/exp_circulation.yaml').SOLVER
        solver = Solver(env, None, cfg)
        pol = env.trainable_policy(cfg.optim, cfg.init_range)
        assert np.allclose(pol.actions_v[0], [0, 0, 0, 0, 0, 0, 0.02, 0.04])          # init_range of the reference config
        pol.actions_v[:] = np.array([0.0, 0.0, 0.0, 0.0, 0.1, 0.0, 0.02, 0.04])
        pol.actions_p[:] = np.array([0.55, 0.5, 0.27, 0.0, 0.0, 0.0, 0.0, 0.0])
        q0 = sf.get_state(0)['q'].copy()
        info, g = solver.forward_backward(te.get_state()['state'], pol, env.horizon, env.horizon_action)
        grads[mode] = (info['loss'], g)
        if mode == 'resident':
            q6 = sf.get_state(6)['q']
            assert q6[:, 16].min() < 0.9 * q0[:, 16].max()                 # the jet cooled part of the slab (low_T = 0)
            assert (q6[:, :15] == q0[:, :15]).all() and (q6[:, 17:] == q0[:, 17:]).all()
            assert g.shape == (7, 8) and np.isfinite(g).all()
            assert np.abs(g[:6, 6]).max() > 0 and np.abs(g[:6, 7]).max() > 0 and np.abs(g[:6, 4]).max() > 0
    assert abs(grads['chunked'][0] - grads['resident'][0]) < 1e-9 * abs(grads['resident'][0])
    assert np.abs(grads['chunked'][1] - grads['resident'][1]).max() < 1e-9 * np.abs(grads['resident'][1]).max()


ICE_MINI = dict(quality=0.5, n_pool=3100, horizon=160, inject_till=300)      # 0.32 s: the first ice cream reaches the cone


def _icecream(engine_lib, target=None, loss=True, **kw):
    return make('IceCreamDynamic-v0', seed=0, loss=loss, target=target, engine_lib=engine_lib, **ICE_MINI, **kw)


def test_icecream_dynamic_env(oracle32):
    """IceCreamDynamic-v0 at a reduced size: BallInjector (stops at inject_till) + Rigid cone with an SDF mesh (collides
    above y = 0.25 only) + plasto-elastic ICECREAM + expanding-range shape-matching loss, record then one Solver pass."""
    env = _icecream(oracle32, loss=False, max_substeps_local=None)
    tgt = Recorder(env).record(write=False)
    used_end = tgt['used'][-1]
    assert used_end.sum() == 300 * 10                                   # 10 particles per substep until inject_till = 300
    x_end = tgt['x'][-1][used_end > 0]
    assert np.isfinite(x_end).all()
    # ice cream that already came down rests on the cone (top near y = 0.3 + ...) instead of the domain floor at 0.05
    landed = x_end[x_end[:, 1] < 0.45]
    assert len(landed) > 200 and landed[:, 1].min() > 0.25
    env = _icecream(oracle32, target=tgt, max_substeps_local=None)
    cfg = load_config('configs/exp_icecream_dynamic.yaml').SOLVER
    pol = env.trainable_policy(cfg.optim, cfg.init_range)
    demo = env.demo_policy()
    pol.actions_v[:] = demo.actions_v; pol.actions_p[:] = demo.actions_p
    env.taichi_env.loss.temporal_range[1] = env.horizon                # evaluate the whole horizon (the curriculum starts at 200)
    info, g = Solver(env, None, cfg).forward_backward(env.taichi_env.get_state()['state'], pol, env.horizon, env.horizon_action)
    assert info['loss'] < 1e-6                                          # the demo policy reproduces its own recording
    pol.actions_v[40:, 0] += 0.0004                                     # drift the cone: the loss wakes up and has a gradient
    info, g = Solver(env, None, cfg).forward_backward(env.taichi_env.get_state()['state'], pol, env.horizon, env.horizon_action)
    assert info['loss'] > 1e-4 and g.shape == (161, 3) and np.isfinite(g).all() and np.abs(g[:160]).max() > 0
    assert pol.trainable[:30].sum() == 0 and pol.trainable[31:-1].all()   # the hold-still prefix of the demo stays frozen


STIR_MINI = dict(quality=0.5, particle_density=6e4, horizon=20)


def _stir(engine_lib, target=None, loss=True, **kw):
    return make('LatteArtStir-v0', seed=0, loss=loss, target=target, engine_lib=engine_lib, **STIR_MINI, **kw)


def test_latteart_stir_env(oracle32):
    """LatteArtStir-v0 at a reduced size: viscous milk on viscous coffee (mu > 0 liquids: the SVD path), a Rigid rod with an SDF
    mesh and softness 100, a loss over every used particle (matching_mat < 0 in the ABI) plus the milk-only figure."""
    env = _stir(oracle32, loss=False, max_substeps_local=None)
    x0 = env.taichi_env.simulator.get_x(0).copy()
    tgt = Recorder(env).record(write=False)
    mat = tgt['mat']
    from fluidlab_amd.configs.macros import COFFEE_VIS, MILK_VIS
    assert set(np.unique(mat)) == {MILK_VIS, COFFEE_VIS}
    moved = np.linalg.norm(tgt['x'][-1] - x0, axis=1)
    near = np.hypot(x0[:, 0] - 0.52, x0[:, 2] - 0.5) < 0.06
    assert moved[near].mean() > 2 * np.median(moved)                     # the rod drags the liquid it passes through
    env = _stir(oracle32, target=tgt, max_substeps_local=None)
    cfg = load_config('configs/exp_latteart_stir.yaml').SOLVER
    pol = env.trainable_policy(cfg.optim, cfg.init_range)
    demo = env.demo_policy()
    pol.actions_v[:] = demo.actions_v; pol.actions_p[:] = demo.actions_p
    pol.actions_v[8:, 2] += 0.001
    env.taichi_env.loss.temporal_range[1] = env.horizon
    info, g = Solver(env, None, cfg).forward_backward(env.taichi_env.get_state()['state'], pol, env.horizon, env.horizon_action)
    assert info['loss'] > info['loss_milk'] > 0                          # all particles vs the milk layer only
    assert g.shape == (21, 3) and np.isfinite(g).all() and np.abs(g[:20, [0, 2]]).max() > 0


ICE_STATIC_MINI = dict(quality=0.5, n_pool=1300, horizon=22, horizon_action=20)


def _icecream_static(engine_lib, target=None, loss=True, **kw):
    return make('IceCreamStatic-v0', seed=0, loss=loss, target=target, engine_lib=engine_lib, **ICE_STATIC_MINI, **kw)


def test_icecream_static_env(oracle32):
    """IceCreamStatic-v0 at a reduced size: a controllable Injector (action gradient through injector.act and move_kernel)
    above a static SDF cone that grid_op collides with."""
    env = _icecream_static(oracle32, loss=False, max_substeps_local=None)
    cone = env.taichi_env.statics[0]
    assert cone.has_dynamics and cone.sdf([[0.5, 0.2, 0.5]])[0] < 0 < cone.sdf([[0.5, 0.5, 0.5]])[0]
    tgt = Recorder(env).record(write=False)
    assert tgt['used'][-1].sum() == 20 * 10 * 6                       # flux 6 per substep while actions last (horizon_action)
    env = _icecream_static(oracle32, target=tgt, max_substeps_local=None)
    cfg = load_config('configs/exp_icecream_static.yaml').SOLVER
    pol = env.trainable_policy(cfg.optim, cfg.init_range)
    demo = env.demo_policy()
    pol.actions_v[:] = demo.actions_v; pol.actions_p[:] = demo.actions_p
    pol.actions_v[5:, 0] += 0.0005
    env.taichi_env.loss.temporal_range[1] = env.horizon
    info, g = Solver(env, None, cfg).forward_backward(env.taichi_env.get_state()['state'], pol, env.horizon, env.horizon_action)
    assert info['loss'] > 0 and g.shape == (21, 3) and np.isfinite(g).all() and np.abs(g).max() > 0


def _gathering(engine_lib, **kw):
    return make('GatheringEasy-v0', seed=0, loss=True, engine_lib=engine_lib, quality=0.5, particle_density=3e4, horizon=12,
                max_substeps_local=None, **kw)


def test_gathering_easy_env(oracle32):
    """GatheringEasy-v0 at a reduced size: MAT_RIGID bodies floating in water, pushed by a Rigid plate (soft SDF contact), L1 loss on
    the bodies' x; the gradient w.r.t. the plate's actions comes back through contact, shape matching and the water."""
    env = _gathering(oracle32)
    te = env.taichi_env
    assert te.simulator.n_bodies == 3
    cfg = load_config('configs/exp_gathering_easy.yaml').SOLVER
    pol = env.trainable_policy(cfg.optim, cfg.init_range)
    assert pol.trainable[:12].all() and pol.status[:12].max() == 0              # the first 50 steps of a sweep are the trainable push
    pol.actions_v[:, 0] = 0.003
    pol.actions_p[:] = [0.46, 0.42, 0.5]                                       # start just left of the bodies... inside the plate's boundary box
    te.loss.temporal_range[1] = env.horizon
    x0 = te.simulator.get_x(0).copy()
    info, g = Solver(env, None, cfg).forward_backward(te.get_state()['state'], pol, env.horizon, env.horizon_action)
    mat = te.simulator.particles_i.mat.to_numpy()
    from fluidlab_amd.configs.macros import RIGID
    x1 = te.get_state()['state']['x']
    assert np.isfinite(x1).all() and info['loss'] > 0
    assert g.shape == (13, 3) and np.isfinite(g).all()
    assert np.abs(g[:12, 0]).max() > 0
    # each body is still rigid
    bid = np.asarray(te.particles['body_id'])
    for b in (1, 2):
        sel = bid == b
        d0 = np.linalg.norm(x0[sel][:, None] - x0[sel][None], axis=2)
        d1 = np.linalg.norm(x1[sel][:, None] - x1[sel][None], axis=2)
        assert np.abs(d1 - d0).max() < 5e-5


# ------------------------------------------------------------------------------------------------------------------------------
# Pouring / Transporting / Mixing / GatheringO: the remaining task environments, at reduced sizes
def _small(name, engine_lib, **kw):
    args = dict(seed=0, loss=True, engine_lib=engine_lib, quality=0.5, particle_density=3e4, horizon=12, max_substeps_local=None)
    args.update(kw)
    return make(name, **args)


def _final_frame(te, horizon):
    sim = te.simulator
    N = sim.n_particles
    x = np.zeros((N, 3), sim.engine.dtype); used = np.zeros((N,), np.int32)
    sim.engine.get_frame(horizon * sim.n_substeps, x=x, used=used)
    return dict(x=x, used=used)


def _solver_pass(env, cfg_file, prepare=None):
    cfg = load_config(cfg_file).SOLVER
    pol = env.trainable_policy(cfg.optim, cfg.init_range)
    if prepare is not None:
        prepare(pol)
    te = env.taichi_env
    info, g = Solver(env, None, cfg).forward_backward(te.get_state()['state'], pol, env.horizon, env.horizon_action)
    return info, g, pol


def test_pouring_env(oracle32):
    """Pouring-v0 reduced: AgentPouring = Rigid glass colliding at particles and nodes + collector; PouringLoss on the host."""
    env = _small('Pouring-v0', oracle32, horizon=10)
    te = env.taichi_env
    assert te.agent.collide_type == 'both' and te.agent.action_dim == 6

    def prepare(pol):
        pol.actions_v[:, 5] = 0.02                      # tilt fast enough to matter within ten steps
    info, g, pol = _solver_pass(env, 'configs/exp_pouring.yaml', prepare)
    assert g.shape == (11, 6) and np.isfinite(g).all() and np.isfinite(info['loss'])
    assert np.abs(g[:10, 5]).max() > 0                  # the trained component: rotation about z
    assert np.isfinite(_final_frame(te, 10)['x']).all()
    # the loss counts the milk's displacement (L1) plus the constant attraction weight per step (pouring_loss.py:150)
    assert info['loss'] > 10 * 1.0
    # optimize() leaves everything but w_z alone
    before = pol.comp_actions.copy()
    pol.optimize(g, info)
    changed = np.abs(pol.comp_actions - before) > 0
    assert changed[:, 5].any() and not changed[:, :5].any()


def test_pouring_collector_in_env(oracle32):
    """the collector box of agent_pouring.yaml starts at y = 0.1: liquid released without a glass falls through it and is taken"""
    env = _small('Pouring-v0', oracle32, horizon=40, loss=False)
    te = env.taichi_env
    pol = env.demo_policy()
    pol.actions_p[:] = [0.2, 0.8, 0.5, 0, 0, 0]        # the glass is elsewhere
    te.apply_agent_action_p(pol.get_actions_p())
    te.simulator.engine.set_frame(0, v=np.tile([0.0, -8.0, 0.0], (te.simulator.n_particles, 1)).astype(np.float32))
    for i in range(40):
        te.step(pol.get_action_v(i))
    st = te.get_state()['state']
    used = np.asarray(st['used'])
    assert 0 < used.sum() < len(used)
    assert (st['x'][used == 0] == -100.0).all()
    assert (st['x'][used == 1][:, 1] >= 0.1 - 2e-3).all()            # at most one substep's fall below the collector face


def test_transporting_env(oracle32):
    """Transporting-v0 reduced: AgentJetBot = turning Injector + WATER collector, a RIGID_HEAVY cube in a z-locked slab,
    TransportingLoss with the pairwise water-cube attraction."""
    env = _small('Transporting-v0', oracle32, horizon=10, n_pool=400, particle_density=2e5)
    te = env.taichi_env
    from fluidlab_amd.configs.macros import RIGID_HEAVY
    mat = te.simulator.particles_i.mat.to_numpy()
    assert (mat == RIGID_HEAVY).sum() > 8 and te.agent.action_dim == 6

    def prepare(pol):
        pol.actions_p[:] = [0.42, 0.5, 0.5, 0.0, 0.0, 0.0]    # right of the cube: the jet leaves the nozzle towards -x ...
        pol.actions_v[:, 5] = 0.002                            # ... while the robot turns slowly
    info, g, pol = _solver_pass(env, 'configs/exp_transporting.yaml', prepare)
    st = _final_frame(te, 10)                                                 # (the backward pass left the simulator at frame 0)
    assert int(st['used'][:400].sum()) == 10 * 10 * 4                         # flux 4 per substep, 10 substeps per step
    assert g.shape == (11, 6) and np.isfinite(g).all()
    assert np.abs(g[:10, 0]).max() > 0 and np.abs(g[:10, 5]).max() > 0      # trained: v_x and w_z
    assert info['attraction_loss'] > 0 and info['dist_loss'] > 0
    assert not pol.trainable[-1] and pol.trainable[:-1].all()
    # z is locked: nothing moves out of its plane
    x0 = te.simulator.get_x(0)
    cube = mat == RIGID_HEAVY
    assert np.abs(st['x'][cube][:, 2] - x0[cube][:, 2]).max() < 1e-4          # (fp32 shape matching: the fitted rotation is about z up to rounding)


def test_transporting_loss_matches_pairwise_sum(oracle32):
    """host_loss.pairwise_l1 against the explicit double loop of transporting_loss.py:94-99"""
    from fluidlab_amd.fluidengine.losses import pairwise_l1
    rng = np.random.RandomState(0)
    a, b = rng.uniform(size=(50, 3)), rng.uniform(size=(7, 3))
    a[4] = b[3]                                                # a tie: |.| has subgradient 0 there (ti.abs' adjoint is sign)
    t, ga, gb = pairwise_l1(a, b)
    ref = sum(np.abs(a[i] - b[j]).sum() for i in range(50) for j in range(7))
    assert abs(t - ref) < 1e-10
    assert np.abs(ga - np.sign(a[:, None] - b[None]).sum(1)).max() == 0
    assert np.abs(gb - np.sign(b[None] - a[:, None]).sum(0)).max() == 0
    t2, g2, _ = pairwise_l1(a)
    assert abs(t2 - np.abs(a[:, None] - a[None]).sum()) < 1e-9 and np.abs(g2 - 2 * np.sign(a[:, None] - a[None]).sum(1)).max() == 0


def test_mixing_env(oracle32):
    """Mixing-v0 reduced: viscous milk block on viscous coffee, stirred by a Rigid rod; MixingLoss (negative pairwise spread)."""
    env = _small('Mixing-v0', oracle32, horizon=10)
    te = env.taichi_env

    def prepare(pol):
        pol.actions_p[:] = [0.5, 0.62, 0.5]                  # the rod's tip in the milk
        pol.actions_v[:, 0] = 0.005
    info, g, pol = _solver_pass(env, 'configs/exp_mixing.yaml', prepare)
    assert info['loss'] < 0                                    # minus a sum of distances
    assert g.shape == (11, 3) and np.isfinite(g).all() and np.abs(g[:10, [0, 2]]).max() > 0
    assert pol.trainable[:10].all() and pol.status[:10].max() == 0
    full = env.trainable_policy(load_config('configs/exp_mixing.yaml').SOLVER.optim, load_config('configs/exp_mixing.yaml').SOLVER.init_range)
    assert full.comp_actions_shape == (11, 3)


def test_latest_pos_follows_the_effector(oracle32):
    """Effector.latest_pos (effector.py:149-151), read by the Gathering / Mixing policies to steer back to a rest pose"""
    env = _small('Mixing-v0', oracle32, horizon=4, loss=False)
    te = env.taichi_env
    te.apply_agent_action_p(np.array([0.5, 0.62, 0.5]))
    assert np.allclose(te.agent.rigid.latest_pos.to_numpy()[0], [0.5, 0.62, 0.5])
    te.step(np.array([0.005, 0.0, -0.002]))
    te.step(np.array([0.005, 0.0, -0.002]))
    # the reference's field holds pos[f] at the START of the last move (effector.py:146-152): one substep short of the current pose
    ns = te.simulator.n_substeps
    last = np.array([0.5, 0.62, 0.5]) + (2 - 1.0 / ns) * np.array([0.005, 0.0, -0.002])
    assert np.allclose(te.agent.rigid.latest_pos.to_numpy()[0], last, atol=1e-6)
    from fluidlab_amd.optimizer.policies import MixingPolicy
    cfg = load_config('configs/exp_mixing.yaml').SOLVER
    pol = MixingPolicy(cfg.optim, cfg.init_range, 3, 100, env.action_range, fix_dim=[1])
    a = pol.get_action_v(50, agent=te.agent, update=True)
    assert np.allclose(a, (np.array([0.5, 0.73, 0.5]) - last) / 30, atol=1e-6)


def test_latteart_stir_policy_freezes_progressively():
    """LatteArtStirPolicy.optimize (policies.py:172-194): the learning rate drops and the early actions are frozen as the loss'
    temporal range grows; only `trainable` is touched, not `freeze_till`."""
    from fluidlab_amd.optimizer.policies import LatteArtStirPolicy
    cfg = load_config('configs/exp_latteart_stir.yaml').SOLVER
    pol = LatteArtStirPolicy(cfg.optim, cfg.init_range, 3, 500, np.array([-0.007, 0.007]), fix_dim=[1])
    g = np.ones(pol.comp_actions_shape)
    before = pol.comp_actions.copy()
    pol.optimize(g, {'temporal_range': 90})
    assert pol.trainable.all() and pol.optim.lr == pol.optim.init_lr
    for tr, frozen, lr in ((120, 0, 1.0), (160, 50, 0.5), (210, 100, 0.5), (260, 150, 0.2), (420, 300, 0.2)):
        pol.optimize(g, {'temporal_range': tr})
        assert not pol.trainable[:frozen].any() and pol.trainable[max(frozen, 300 if tr > 400 else frozen):].all(), tr
        assert np.isclose(pol.optim.lr, pol.optim.init_lr * lr) and pol.freeze_till == 0
    assert not np.array_equal(pol.comp_actions[300:], before[300:])           # (frozen rows still coast on Adam's momentum, as in the reference)


def test_mixing_policy_cycle():
    """MixingPolicy (policies.py:306-338): 50 trainable steps, 30 steps back to the rest pose, per 80-step cycle"""
    from fluidlab_amd.optimizer.policies import MixingPolicy
    cfg = load_config('configs/exp_mixing.yaml').SOLVER
    pol = MixingPolicy(cfg.optim, cfg.init_range, 3, 200, np.array([-0.007, 0.007]), fix_dim=[1])
    assert pol.trainable[:50].all() and not pol.trainable[50:80].any() and pol.trainable[80:130].all()

    class _Pos:
        def to_numpy(self):
            return np.array([[0.4, 0.6, 0.5]])

    class _Agent:
        class rigid:
            latest_pos = _Pos()
    a = pol.get_action_v(50, agent=_Agent, update=True)
    assert np.allclose(a, (np.array([0.5, 0.73, 0.5]) - [0.4, 0.6, 0.5]) / 30)
    g = np.ones((201, 3))
    pol.optimize(g, {'temporal_range': 170})
    assert pol.freeze_till == 10 and not pol.trainable[:10].any()


def test_gathering_o_env(oracle32):
    """GatheringO-v0 reduced: the O-tank's island as a static SDF collider in grid_op + Rigid plate + two rigid bodies;
    squared-distance loss to the goal in the xz plane."""
    env = _small('GatheringO-v0', oracle32, horizon=12)
    te = env.taichi_env
    assert te.simulator.n_bodies == 3 and len(te.statics.statics) == 1

    def prepare(pol):
        pol.actions_v[:, 0] = 0.003
    te.loss.temporal_range[1] = env.horizon
    info, g, pol = _solver_pass(env, 'configs/exp_gatheringO.yaml', prepare)
    assert info['loss'] > 0 and g.shape == (13, 3) and np.isfinite(g).all()
    assert np.abs(g[:12, 0]).max() > 0
    # water that started outside the island does not enter it (like the reference, the water block is sampled over the whole
    # tank, island included: gatheringo_env.py:54-59)
    st = _final_frame(te, 12)
    mat = te.simulator.particles_i.mat.to_numpy()
    from fluidlab_amd.configs.macros import WATER
    island = te.statics.statics[0]
    outside0 = island.sdf(te.simulator.get_x(0)[mat == WATER]) > 0.005
    assert outside0.sum() > 1000 and (island.sdf(st['x'][mat == WATER])[outside0] > -0.01).all()


def test_optional_dt_keeps_the_reference_relation_between_dt_and_substeps(oracle32):
    """The reference fixes dt = 2e-4 and n_substeps = int(2e-3 / dt) (mpm:24, 30).  `dt` is this repository's addition for BASELINE config 5's 256^3
    grid (the stiff materials are past their Courant limit there at 2e-4): the default is the reference's, an override keeps 2e-3 of simulated time
    per step, and the injector dispenses `flux` particles per SUBSTEP either way (injector.py:80-105)."""
    env = _icecream(oracle32, loss=False, max_substeps_local=None)
    assert env.taichi_env.simulator.dt == 2e-4 and env.taichi_env.simulator.n_substeps == 10
    kw = dict(ICE_MINI, horizon=4, inject_till=10**9)
    env = make('IceCreamDynamic-v0', seed=0, loss=False, engine_lib=oracle32, max_substeps_local=40, ckpt_dest='cpu', dt=5e-5, **kw)
    te = env.taichi_env
    assert te.simulator.dt == 5e-5 and te.simulator.n_substeps == 40
    pol = env.demo_policy()
    te.apply_agent_action_p(pol.get_actions_p())
    for i in range(2):
        te.step(pol.get_action_v(i))
    assert te.simulator.get_used().sum() == 2 * 40 * 10


def test_work_stats_counted_form(oracle32):
    """fe_get_work_stats_n writes min(n, FE_WORK_STATS) entries (the list grew under one symbol name); the oracle has no work lists: zeros"""
    import ctypes as C
    import scenarios as S
    eng = S.make_engine(oracle32, S.water_block(n_grid=8, n_particles=64))
    out = (C.c_longlong * 6)(*([7] * 6))
    assert oracle32.lib.fe_get_work_stats_n(eng.h, 0, out, 4) == 0
    assert list(out) == [0, 0, 0, 0, 7, 7]
    ws = eng.get_work_stats(0)
    assert ws['n_items'] == 0 and ws['n_split9_waves'] == 0 and ws['n_split3_waves'] == 0
