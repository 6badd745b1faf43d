"""Generates tests/golden/host_golden.npz from the REFERENCE's own host-side (pure NumPy) code.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_host.py

The simulation kernels of the reference are Taichi and cannot run here, but three pieces of the path's host
logic are plain NumPy and can: particle sampling (fluidengine/bodies/bodies.py), the Adam update
(optimizer/optim.py) and LatteArt's scripted demo policy (envs/latteart_env.py:113-140).  They are imported
from /root/reference with the absent third-party packages (taichi, trimesh, gym, ...) replaced by inert
placeholders that are never called on these code paths; outputs are stored as data only."""
import sys
import types
from unittest.mock import MagicMock

import numpy as np

REF = '/root/reference'
sys.path.insert(0, REF)
for name in ['taichi', 'trimesh', 'mesh_to_sdf', 'cv2', 'yacs', 'yacs.config', 'skimage', 'pynput', 'matplotlib', 'matplotlib.pyplot',
             'sklearn.neighbors', 'torch.utils.tensorboard', 'pyrender', 'OpenGL', 'OpenGL.GL',
             'fluidlab.fluidengine.renderers.gl_renderer_src', 'fluidlab.fluidengine.renderers.gl_renderer_src.flex_renderer']:
    sys.modules.setdefault(name, MagicMock())
gym = types.ModuleType('gym')
gym.Env = object
gym.spaces = types.ModuleType('gym.spaces')
gym.spaces.Box = MagicMock()
gym.envs = MagicMock()
gym.register = lambda *a, **k: None
sys.modules['gym'] = gym
sys.modules['gym.spaces'] = gym.spaces
sys.modules['gym.envs'] = MagicMock()
sys.modules['gym.envs.registration'] = MagicMock()

out = {}

# ---- 1. particle sampling --------------------------------------------------------------------------
from fluidlab.fluidengine.bodies.bodies import Bodies          # noqa: E402
from fluidlab.configs.macros import MILK, COFFEE, WATER, ICECREAM  # noqa: E402


def summarize(tag, p):
    x = p['x']
    out[f'{tag}_n'] = np.array(len(x))
    out[f'{tag}_head'] = x[:64]
    out[f'{tag}_tail'] = x[-64:]
    out[f'{tag}_sum'] = x.sum(0)
    out[f'{tag}_sqsum'] = (x * x).sum(0)
    out[f'{tag}_used_sum'] = np.array(int(np.sum(p['used'])))
    out[f'{tag}_mat_sum'] = np.array(int(np.sum(p['mat'])))
    out[f'{tag}_rho_sum'] = np.array(float(np.sum(p['rho'])))
    out[f'{tag}_body_n'] = np.array(p['bodies']['n_particles'])


b = Bodies(dim=3, particle_density=1e6)                         # LatteArt-v0, latteart_env.py:54-66
b.add_body(type='nowhere', n_particles=60000, material=MILK)
b.add_body(type='cylinder', center=(0.5, 0.55, 0.5), height=0.1, radius=0.42, material=COFFEE)
summarize('latte', b.get())

b = Bodies(dim=3, particle_density=2e5)                         # all samplers / fillings
b.add_body(type='cube', lower=(0.2, 0.2, 0.2), upper=(0.4, 0.4, 0.4), material=WATER)
b.add_body(type='ball', center=(0.6, 0.3, 0.6), radius=0.1, material=WATER)
b.add_body(type='cube', lower=(0.5, 0.5, 0.5), size=(0.1, 0.2, 0.1), material=ICECREAM, filling='grid', euler=(0.0, 30.0, 10.0))
b.add_body(type='cylinder', center=(0.3, 0.7, 0.3), height=0.1, radius=0.08, material=WATER, filling='natural')
b.add_body(type='ball', center=(0.7, 0.7, 0.3), radius=0.06, material=WATER, filling='natural')
summarize('mix', b.get())

# ---- 2. Adam ---------------------------------------------------------------------------------------
from fluidlab.optimizer.optim import Adam                        # noqa: E402
cfg = types.SimpleNamespace(lr=1e-3, beta_1=0.9, beta_2=0.99, epsilon=1e-8)
opt = Adam((7, 3), cfg)
rng = np.random.RandomState(11)
params = rng.normal(size=(7, 3))
grads = rng.normal(size=(5, 7, 3)) * np.array([1.0, 1e-3, 1e3])
traj = []
for g in grads:
    params = opt.step(params, g)
    traj.append(params.copy())
out['adam_params0_seed'] = np.array(11)
out['adam_traj'] = np.array(traj)

# ---- 3. LatteArt demo policy -----------------------------------------------------------------------
from fluidlab.envs.latteart_env import LatteArtEnv                # noqa: E402
fake = types.SimpleNamespace(horizon_action=250, agent=types.SimpleNamespace(action_dim=3))
pol = LatteArtEnv.demo_policy(fake, user_input=False)
out['latte_demo_actions_v'] = np.asarray(pol.actions_v)
out['latte_demo_actions_p'] = np.asarray(pol.actions_p)

# ---- 4. pose conventions: NumPy helpers of utils/geom.py, Mesh.init_transform (mesh.py:97-103), Effector.init_rot (effector.py:45)
import fluidlab.utils.geom as geom_utils                          # noqa: E402
from scipy.spatial.transform import Rotation                       # noqa: E402
rng = np.random.RandomState(3)
poses = []
for _ in range(6):
    pos, euler, scale = rng.uniform(-0.5, 1.0, 3), rng.uniform(-180, 180, 3), rng.uniform(0.2, 1.5, 3)
    quat = geom_utils.xyzw_to_wxyz(Rotation.from_euler('zyx', euler[::-1], degrees=True).as_quat())
    T_init = geom_utils.trans_quat_to_T(pos, quat) @ geom_utils.scale_to_T(scale)
    pts = rng.normal(size=(5, 3))
    poses.append(np.concatenate([pos, euler, scale, quat, T_init.ravel(), geom_utils.transform_by_T_np(pts, T_init).ravel(), pts.ravel(),
                                 geom_utils.transform_by_quat_np(pts, quat).ravel()]))
out['pose_cases'] = np.array(poses)

# ---- 5. normalize_mesh (utils/mesh.py:33-46) on bare vertex arrays
from fluidlab.utils.mesh import normalize_mesh                     # noqa: E402


class _M:
    def __init__(self, v):
        self.vertices = np.array(v, dtype=np.float64)

    def copy(self):
        return _M(self.vertices.copy())


va, vb = rng.normal(size=(40, 3)) * [1.0, 3.0, 0.5] + [2.0, -1.0, 0.3], rng.normal(size=(25, 3)) * 2.0
out['normalize_in_a'], out['normalize_in_b'] = va, vb
out['normalize_self'] = normalize_mesh(_M(va)).vertices
out['normalize_by_other'] = normalize_mesh(_M(va), _M(vb)).vertices

# ---- 6. material tables (configs/macros.py)
import fluidlab.configs.macros as RM                               # noqa: E402
names = ['WATER', 'MILK', 'COFFEE', 'ELASTIC', 'ICECREAM', 'RIGID', 'RIGID_HEAVY', 'RIGID_LIGHT', 'MILK_VIS', 'COFFEE_VIS']
out['mat_names'] = np.array(names)
out['mat_ids'] = np.array([getattr(RM, n) for n in names])
out['mat_table'] = np.array([[RM.MU[getattr(RM, n)], RM.LAMDA[getattr(RM, n)], RM.RHO[getattr(RM, n)], RM.MAT_CLASS[getattr(RM, n)]] for n in names], dtype=np.float64)
fr = ['CUP', 'TANK', 'BOWL', 'LADDLE', 'CONE', 'BOTTLE', 'PILLAR', 'STIRRER', 'PLATE']
out['friction_names'] = np.array(fr)
out['friction'] = np.array([RM.FRICTION[getattr(RM, n)] for n in fr], dtype=np.float64)
out['friction_ids'] = np.array([getattr(RM, n) for n in fr])
out['nowhere'] = np.array(RM.NOWHERE, dtype=np.float64)
out['eps'] = np.array(RM.EPS, dtype=np.float64)

# ---- 7. the staged policies (optimizer/policies.py): which steps are trainable, which stage each step is in
from fluidlab.optimizer.policies import GatheringPolicy, GatheringOPolicy, MixingPolicy, TransportingPolicy, IceCreamDynamicPolicy, IceCreamStaticPolicy  # noqa: E402
optim_cfg = types.SimpleNamespace(type='Adam', lr=1e-3, beta_1=0.9, beta_2=0.99, epsilon=1e-8)
for cls, dim, H in ((GatheringPolicy, 3, 300), (GatheringOPolicy, 3, 300), (MixingPolicy, 3, 200), (TransportingPolicy, 6, 50),
                    (IceCreamDynamicPolicy, 3, 900), (IceCreamStaticPolicy, 3, 50)):
    ir = types.SimpleNamespace(v=(np.zeros(dim), np.zeros(dim)), p=(np.full(dim, 0.5), np.full(dim, 0.5)))
    pol = cls(optim_cfg, ir, dim, H, np.array([-0.01, 0.01]), fix_dim=[1])
    out[f'policy_{cls.__name__}_trainable'] = np.asarray(pol.trainable)
    if hasattr(pol, 'status'):
        out[f'policy_{cls.__name__}_status'] = np.asarray(pol.status)
# one optimisation step of the base class: masks (trainable rows, fixed columns) then Adam then the clip of the velocity rows
pol = TransportingPolicy(optim_cfg, types.SimpleNamespace(v=(np.zeros(6), np.zeros(6)), p=(np.full(6, 0.5), np.full(6, 0.5))), 6, 8,
                         np.array([-0.0005, 0.0005]), fix_dim=[1, 2, 3, 4])
g = np.random.RandomState(4).normal(size=(9, 6))
pol.optimize(g, {'temporal_range': 8})
out['policy_step_grads'] = g
out['policy_step_result'] = np.vstack([pol.actions_v, pol.actions_p[None]])

# ---- 8. bodies of the later envs: Transporting's rotated 'natural' cube, Mixing's block (bodies.py)
b = Bodies(dim=3, particle_density=1e6)
b.add_body(type='nowhere', n_particles=1000, material=WATER)
b.add_body(type='cube', lower=(0.275, 0.475, 0.475), size=(0.05, 0.05, 0.05), euler=(45.0, 45.0, 45.0), color=(1.0, 0.5, 0.5, 1.0), filling='natural',
           material=RM.RIGID_HEAVY)
b.add_body(type='cube', lower=(0.425, 0.55, 0.425), upper=(0.575, 0.7, 0.575), material=RM.MILK_VIS)
summarize('later', b.get())

np.savez_compressed('/root/repo/tests/golden/host_golden.npz', **out)
print({k: getattr(v, 'shape', None) for k, v in out.items()})
