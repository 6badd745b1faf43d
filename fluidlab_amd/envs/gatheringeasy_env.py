"""GatheringEasy-v0 (fluidlab/envs/gatheringeasy_env.py): a Rigid plate sweeps through a water tank to herd two floating
MAT_RIGID bodies towards x = 0.8.  Everything this round added meets here: shape-matching rigid bodies in water, a moving SDF
collider with softness, its pose adjoint and an L1 loss on the bodies' particles.

The reference's ducks are mesh bodies (trimesh voxelisation, unavailable here): two rigid boxes stand in, at the ducks' places.
The plate's collision mesh is an analytic thin box.  `quality`, `particle_density`, `horizon` scale the scene for tests."""
import os

import numpy as np

from fluidlab_amd.configs.macros import RIGID, TANK, WATER
from fluidlab_amd.fluidengine.losses import GatheringEasyLoss
from fluidlab_amd.fluidengine.meshes import sdf_box
from fluidlab_amd.fluidengine.taichi_env import TaichiEnv
from fluidlab_amd.optimizer.policies import GatheringPolicy
from fluidlab_amd.utils.config import CfgNode
from fluidlab_amd.utils.mesh import get_raw_mesh_path
from fluidlab_amd.utils.misc import get_cfg_path
from .fluid_env import FluidEnv


class GatheringEasyEnv(FluidEnv):
    def __init__(self, version=0, loss=True, loss_type='diff', seed=None, renderer_type=None, quality=1, particle_density=1e6, horizon=840,
                 max_substeps_local=50, ckpt_dest='disk', engine_lib=None, device=0):
        if seed is not None:
            self.seed(seed)
        self.horizon = horizon
        self.horizon_action = horizon
        self.target_file = None
        self._n_obs_ptcls_per_body = 500
        self.loss = loss
        self.loss_type = loss_type
        self.action_range = np.array([-0.003, 0.003])
        self.taichi_env = TaichiEnv(dim=3, quality=quality, particle_density=particle_density, max_substeps_local=max_substeps_local,
                                    gravity=(0.0, -20.0, 0.0), horizon=self.horizon, ckpt_dest=ckpt_dest, engine_lib=engine_lib, device=device)
        self.build_env()
        self.gym_misc()

    def setup_agent(self):
        agent_cfg = CfgNode()
        agent_cfg.merge_from_file(get_cfg_path('agent_gatheringeasy.yaml'))
        # plate.obj stand-in: the real mesh, normalised, has half extents (0.333, 0.5, 0.083) -- thin along its z, which the yaml's
        # euler (0, 90, 0) turns into the world x the plate pushes along
        agent_cfg.effectors[0]['mesh']['sdf'] = sdf_box((0.333, 0.5, 0.083))
        agent_cfg.effectors[0]['mesh']['sdf_res'] = 64
        self.taichi_env.setup_agent(agent_cfg)
        self.agent = self.taichi_env.agent

    def setup_statics(self):
        self.taichi_env.add_static(file='tank.obj', pos=(0.5, 0.4, 0.5), euler=(0.0, 0.0, 0.0), scale=(1.0, 0.92, 0.92), material=TANK,
                                   has_dynamics=False)

    def setup_bodies(self):
        self.taichi_env.add_body(type='cube', lower=(0.05, 0.3, 0.17), upper=(0.95, 0.45, 0.83), material=WATER)
        if os.path.exists(get_raw_mesh_path('duck.obj')):            # the asset tree has the ducks: the reference's mesh bodies (:61-80)
            self.taichi_env.add_body(type='mesh', file='duck.obj', pos=(0.22, 0.5, 0.45), scale=(0.10, 0.10, 0.10), euler=(0, -75.0, 0.0),
                                     color=(1.0, 1.0, 0.3, 1.0), filling='grid', material=RIGID)
            self.taichi_env.add_body(type='mesh', file='duck.obj', pos=(0.28, 0.5, 0.57), scale=(0.10, 0.10, 0.10), euler=(0, -95.0, 0.0),
                                     color=(1.0, 0.5, 0.5, 1.0), filling='grid', material=RIGID)
            return
        # duck.obj stand-ins: boxes with three different edge lengths.  (A ball's covariance H is isotropic, its singular values
        # coincide, and the reference's SVD adjoint 1/clamp(s_j^2 - s_i^2) (mpm:272-292) then amplifies rounding noise: fp32 and fp64
        # runs of the same code disagree in the gradient.)
        self.taichi_env.add_body(type='cube', lower=(0.18, 0.47, 0.42), upper=(0.26, 0.52, 0.48), material=RIGID)
        self.taichi_env.add_body(type='cube', lower=(0.25, 0.47, 0.53), upper=(0.31, 0.53, 0.62), material=RIGID)

    def setup_boundary(self):
        self.taichi_env.setup_boundary(type='cube', lower=(0.06, 0.3, 0.18), upper=(0.94, 0.95, 0.82))

    def setup_loss(self):
        self.taichi_env.setup_loss(loss_cls=GatheringEasyLoss, type=self.loss_type, matching_mat=RIGID, weights={'dist': 1.0})

    def trainable_policy(self, optim_cfg, init_range):
        return GatheringPolicy(optim_cfg, init_range, self.agent.action_dim, self.horizon_action, self.action_range, fix_dim=[1])
