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

np.savez_compressed('/root/repo/tests/golden/host_golden.npz', **out)
print({k: getattr(v, 'shape', None) for k, v in out.items()})
