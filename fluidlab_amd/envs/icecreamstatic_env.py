"""IceCreamStatic-v0 (fluidlab/envs/icecreamstatic_env.py): a controllable Injector squeezes plasto-elastic ice cream onto a
*static* cone (an SDF collider in grid_op); the loss matches the ICECREAM1 particles to a recorded swirl.

`quality`, `n_pool`, `horizon`, `horizon_action` scale the scene for tests; defaults are the reference's (64^3, 100k pool, 550
steps of which 500 carry actions).  The cone's collision mesh is an analytic stand-in (fluidengine/meshes.py: sdf_cone_tip)."""
import os

import numpy as np

from fluidlab_amd.configs.macros import CONE, ICECREAM1
from fluidlab_amd.fluidengine.losses import IceCreamStaticLoss
from fluidlab_amd.fluidengine.meshes import sdf_cone_tip
from fluidlab_amd.fluidengine.taichi_env import TaichiEnv
from fluidlab_amd.optimizer.policies import ActionsPolicy, IceCreamStaticPolicy
from fluidlab_amd.utils.config import CfgNode
from fluidlab_amd.utils.misc import get_cfg_path, get_tgt_path
from .fluid_env import FluidEnv


class IceCreamStaticEnv(FluidEnv):
    def __init__(self, version=0, loss=True, loss_type='diff', seed=None, renderer_type=None, quality=1, n_pool=100000, horizon=550,
                 horizon_action=500, max_substeps_local=20, ckpt_dest='disk', target=None, engine_lib=None, device=0):
        if seed is not None:
            self.seed(seed)
        self.horizon = horizon
        self.horizon_action = horizon_action
        self.target_file = get_tgt_path('IceCreamStatic-v0.pkl')
        self._target = target
        self._n_obs_ptcls_per_body = 2000
        self._n_pool = n_pool
        self.loss = loss
        self.loss_type = loss_type
        self.action_range = np.array([-0.005, 0.005])
        self.taichi_env = TaichiEnv(dim=3, quality=quality, particle_density=1e6, max_substeps_local=max_substeps_local,
                                    gravity=(0.0, -5.0, 0.0), horizon=self.horizon, ckpt_dest=ckpt_dest, engine_lib=engine_lib, device=device)
        self.build_env()
        self.gym_misc()

    def setup_agent(self):
        agent_cfg = CfgNode()
        agent_cfg.merge_from_file(get_cfg_path('agent_icecreamstatic.yaml'))
        self.taichi_env.setup_agent(agent_cfg)
        self.agent = self.taichi_env.agent

    def setup_statics(self):
        self.taichi_env.add_static(file='cone.obj', pos=(0.5, 0.1, 0.5), euler=(-90.0, 0.0, 30.0), scale=(0.435, 0.435, 0.435), material=CONE,
                                   has_dynamics=True, sdf=sdf_cone_tip(r_bottom=0.05, r_top=0.3, z_bottom=-0.2, z_top=0.45, dent=0.06), sdf_res=64)

    def setup_bodies(self):
        self.taichi_env.add_body(type='nowhere', n_particles=self._n_pool, material=ICECREAM1)

    def setup_boundary(self):
        self.taichi_env.setup_boundary(type='cube', lower=(0.05, 0.05, 0.05), upper=(0.95, 0.95, 0.95))

    def setup_loss(self):
        target = self._target if self._target is not None else (self.target_file if os.path.exists(self.target_file) else None)
        self.taichi_env.setup_loss(loss_cls=IceCreamStaticLoss, type=self.loss_type, target_file=target, weights={'chamfer': 1.0})

    def demo_policy(self, user_input=False):
        """icecreamstatic_env.py:104-138: a rising spiral that tightens from radius 0.12 to 0.01 over 700 steps (scaled with the
        action horizon)."""
        if user_input:
            raise NotImplementedError
        H = self.horizon_action
        comp_actions_p = np.zeros((1, self.agent.action_dim))
        comp_actions_v = np.zeros((H, self.agent.action_dim))
        init_center = np.array([0.5, 0.36, 0.5])
        y_range, init_radius, final_radius = 0.26, 0.12, 0.01
        init_p = init_center + np.array([init_radius, 0, 0])
        current_p = np.array(init_p)
        horizon_1 = 700 * H / 500
        init_rad_v = 0.01 * 500 / H
        final_rad_v = init_rad_v * init_radius / final_radius
        theta = 0.0
        for i in range(H):
            rad_v = (final_rad_v - init_rad_v) * i / horizon_1 + init_rad_v
            theta += rad_v
            r = i / horizon_1 * (final_radius - init_radius) + init_radius
            target_p = np.array([init_center[0] + r * np.cos(theta), init_center[1] + y_range * i / horizon_1, init_center[2] + r * np.sin(theta)])
            comp_actions_v[i] = target_p - current_p
            current_p += comp_actions_v[i]
        comp_actions_p[0] = init_p
        return ActionsPolicy(np.vstack([comp_actions_v, comp_actions_p]))

    def trainable_policy(self, optim_cfg, init_range):
        return IceCreamStaticPolicy(optim_cfg, init_range, self.agent.action_dim, self.horizon_action, self.action_range, fix_dim=None)
