"""Circulation-v0 -- indoor air circulation (fluidlab/envs/circulation_env.py): an AirCon steers a smoke / temperature
field through a room; the loss reads the temperature at fifteen detector cells.

The room (`room.obj` in the reference) is needed as an SDF and mesh -> SDF conversion is not available here
(fluidengine/meshes.py), so it is an analytic stand-in in the same pose: outer walls plus two partition walls that leave
doorways.  `res`, `horizon`, `solver_iters`, `max_substeps_local` scale the scene for tests; the defaults are the
reference's (128^3 smoke grid, 50 Jacobi sweeps, 1000 steps, checkpoint window of 100 substeps on the host)."""
import numpy as np

from fluidlab_amd.configs.macros import PILLAR, WATER
from fluidlab_amd.fluidengine.losses import CirculationLoss
from fluidlab_amd.fluidengine.taichi_env import TaichiEnv
from fluidlab_amd.optimizer.policies import ActionsPolicy, CirculationPolicy
from fluidlab_amd.utils.config import CfgNode
from fluidlab_amd.utils.misc import get_cfg_path
from .fluid_env import FluidEnv


ROOM_SCALE = np.array([1.4, 3.0, 1.4])          # circulation_env.py's add_static(room.obj, scale=...), about (0.5, 0.5, 0.5)


def sdf_room():
    """Stand-in for room.obj, after the floor plan of the reference's mesh (mapped with fe_mesh_sdf on the asset): the free space
    is x in [0.05, 0.95], z in [0.22, 0.80] over the whole height, split into four rooms -- a wall at x = 0.38 with doorways at
    z in [0.29, 0.38] and [0.59, 0.71], a wall at z = 0.49 between the two rooms on its low-x side, and a wall at x = 0.68 with a
    doorway at z in [0.41, 0.56].  Written in world coordinates, returned in the mesh frame the SDF lattice samples (distances
    divided by the xz scale; only sign and direction matter to the colliders)."""
    def box(w, lo, hi):
        c, h = (np.asarray(lo) + np.asarray(hi)) / 2, (np.asarray(hi) - np.asarray(lo)) / 2
        q = np.abs(w - c) - h
        return np.linalg.norm(np.maximum(q, 0), axis=1) + np.minimum(q.max(axis=1), 0)

    walls = [((0.36, -1, 0.22), (0.40, 2, 0.29)), ((0.36, -1, 0.38), (0.40, 2, 0.59)), ((0.36, -1, 0.71), (0.40, 2, 0.80)),
             ((0.05, -1, 0.47), (0.38, 2, 0.51)),
             ((0.66, -1, 0.22), (0.70, 2, 0.41)), ((0.66, -1, 0.56), (0.70, 2, 0.80))]

    def fn(p):
        w = p * ROOM_SCALE + 0.5
        d = -box(w, (0.05, -1, 0.22), (0.95, 2, 0.80))              # solid outside the free space
        for lo, hi in walls:
            d = np.minimum(d, box(w, lo, hi))
        return d / ROOM_SCALE[0]
    return fn


class CirculationEnv(FluidEnv):
    def __init__(self, version=0, loss=True, loss_type='diff', seed=None, renderer_type=None, res=128, horizon=1000, solver_iters=50,
                 max_substeps_local=100, ckpt_dest='cpu', detectors=None, engine_lib=None, device=0):
        if seed is not None:
            self.seed(seed)
        self.horizon = horizon
        self.horizon_action = horizon
        self.target_file = None
        self._n_obs_ptcls_per_body = 200
        self.loss = loss
        self.loss_type = loss_type
        self.action_range = np.array([-0.1, 0.1])
        self._res, self._iters, self._detectors = res, solver_iters, detectors
        self.taichi_env = TaichiEnv(dim=3, particle_density=1e6, max_substeps_local=max_substeps_local, gravity=(0.0, -20.0, 0.0),
                                    horizon=self.horizon, ckpt_dest=ckpt_dest, engine_lib=engine_lib, device=device)
        self.build_env()
        self.gym_misc()

    def setup_agent(self):
        agent_cfg = CfgNode()
        agent_cfg.merge_from_file(get_cfg_path('agent_circulation.yaml'))
        self.taichi_env.setup_agent(agent_cfg)
        self.agent = self.taichi_env.agent

    def setup_statics(self):
        self.taichi_env.add_static(file='room.obj', pos=(0.5, 0.5, 0.5), euler=(0.0, 0.0, 0.0), scale=(1.4, 3.0, 1.4), material=PILLAR,
                                   sdf_res=min(128, self._res), has_dynamics=True, sdf=sdf_room())

    def setup_bodies(self):
        self.taichi_env.add_body(type='nowhere', n_particles=10, material=WATER)

    def setup_smoke_field(self):
        self.taichi_env.setup_smoke_field(res=self._res, dt=0.03, solver_iters=self._iters, decay=0.99, q_dim=1)
        if self._res != 128:                      # the free slab 60 < j < 68 (smoke_field.py:25-26) scaled with the grid
            sf = self.taichi_env.smoke_field
            sf.lower_y, sf.higher_y = int(round(60 * self._res / 128)), int(round(68 * self._res / 128))

    def setup_boundary(self):
        pass

    def setup_loss(self):
        self.taichi_env.setup_loss(loss_cls=CirculationLoss, type=self.loss_type, weights={'temp': 1.0}, detectors=self._detectors)

    def _get_obs(self):
        state = self.taichi_env.get_state_RL()
        obs = [state['agent'][0].flatten()] if 'agent' in state else []
        if 'smoke_field' in state:                                  # fluid_env.py:120-122
            sf = self.taichi_env.smoke_field
            obs.append(state['smoke_field']['v'][::10, sf.lower_y:sf.higher_y, ::10].flatten())
            obs.append(state['smoke_field']['q'][::10, sf.lower_y:sf.higher_y, ::10].flatten())
        return np.concatenate(obs)

    def demo_policy(self):
        comp_actions_p = np.zeros((1, self.agent.action_dim))
        comp_actions_v = np.zeros((self.horizon_action, self.agent.action_dim))
        comp_actions_p[0] = np.array([0.55, 0.5, 0.27, 0.0, 0.0, 0.0, 0.0, 0.0])
        comp_actions_v[:] = np.array([0.0, 0.0, 0.0, 0.0, 0.1, 0.0, 0.02, 0.04])
        return ActionsPolicy(np.vstack([comp_actions_v, comp_actions_p]))

    def trainable_policy(self, optim_cfg, init_range):
        return CirculationPolicy(optim_cfg, init_range, self.agent.action_dim, self.horizon_action, self.action_range,
                                 fix_dim=[0, 1, 2, 3, 5, 6, 7])
