"""LatteArt-v0 (fluidlab/envs/latteart_env.py): pour milk into coffee to match a recorded pattern.

`quality` / `particle_density` / `n_pool` scale the scene to BASELINE config 3 (128^3, ~200k particles);
the defaults are the reference scene (64^3, 55,480 coffee + 60,000 pool)."""
import numpy as np

from fluidlab_amd.configs.macros import COFFEE, CUP, MILK
from fluidlab_amd.fluidengine.losses import LatteArtLoss
from fluidlab_amd.fluidengine.taichi_env import TaichiEnv
from fluidlab_amd.optimizer.policies import ActionsPolicy, LatteArtPolicy
from fluidlab_amd.utils.config import CfgNode
from fluidlab_amd.utils.misc import get_cfg_path, get_tgt_path
from .fluid_env import FluidEnv


class LatteArtEnv(FluidEnv):
    def __init__(self, version=0, loss=True, loss_type='diff', seed=None, renderer_type=None, quality=1, particle_density=1e6,
                 n_pool=60000, horizon=330, horizon_action=250, max_substeps_local=None, ckpt_dest='disk', target=None,
                 engine_lib=None, device=0):
        if seed is not None:
            self.seed(seed)
        self.horizon = horizon
        self.horizon_action = horizon_action
        self.target_file = get_tgt_path('LatteArt-v0.pkl')
        self._target = target                     # in-memory target (dict) instead of the pickle
        self._n_obs_ptcls_per_body = 1000
        self._n_pool = n_pool
        self.loss = loss
        self.loss_type = loss_type
        self.action_range = np.array([-0.05, 0.05])
        # the reference passes max_substeps_local=50 (latteart_env.py:31); None keeps the whole trajectory in HBM
        self.taichi_env = TaichiEnv(dim=3, quality=quality, particle_density=particle_density, max_substeps_local=max_substeps_local,
                                    gravity=(0.0, -20.0, 0.0), horizon=self.horizon, ckpt_dest=ckpt_dest, engine_lib=engine_lib,
                                    device=device)
        self.build_env()
        self.gym_misc()

    def setup_agent(self):
        agent_cfg = CfgNode()
        agent_cfg.merge_from_file(get_cfg_path('agent_latteart.yaml'))
        self.taichi_env.setup_agent(agent_cfg)
        self.agent = self.taichi_env.agent

    def setup_statics(self):
        self.taichi_env.add_static(file='cup.obj', pos=(0.63, 0.42, 0.5), euler=(0.0, 0.0, 0.0), scale=(1.2, 1.2, 1.2),
                                   material=CUP, has_dynamics=False)

    def setup_bodies(self):
        self.taichi_env.add_body(type='nowhere', n_particles=self._n_pool, material=MILK)
        self.taichi_env.add_body(type='cylinder', center=(0.5, 0.55, 0.5), height=0.1, radius=0.42, material=COFFEE)

    def setup_boundary(self):
        self.taichi_env.setup_boundary(type='cylinder', xz_radius=0.42, xz_center=(0.5, 0.5), y_range=(0.5, 0.95))

    def setup_loss(self):
        import os
        target = self._target if self._target is not None else (self.target_file if os.path.exists(self.target_file) else None)
        # without a recorded target the loss is built empty: call taichi_env.loss.set_target(...) before solving
        self.taichi_env.setup_loss(loss_cls=LatteArtLoss, type=self.loss_type, target_file=target, weights={'chamfer': 1.0})

    def demo_policy(self, user_input=False):
        """The scripted sine pour of latteart_env.py:113-140."""
        assert not user_input, 'interactive policies need a display'
        dim = self.agent.action_dim
        comp_actions_v = np.zeros((self.horizon_action, dim))
        init_p = np.array([0.15, 0.65, 0.5])
        x_range, cycles = 0.7, 3
        amp_range = np.array([0.15, 0.25])
        current_p = np.array(init_p)
        for i in range(self.horizon_action):
            t = (i + 1) / self.horizon_action
            amp = amp_range[1] - np.abs(2 * t - 1) * (amp_range[1] - amp_range[0])
            target_p = np.array([init_p[0] + t * x_range, init_p[1], np.sin(t * (np.pi * 2) * cycles) * amp + 0.5])
            comp_actions_v[i] = target_p - current_p
            current_p += comp_actions_v[i]
        return ActionsPolicy(np.vstack([comp_actions_v, init_p[None, :]]))

    def trainable_policy(self, optim_cfg, init_range):
        return LatteArtPolicy(optim_cfg, init_range, self.agent.action_dim, self.horizon_action, self.action_range)
