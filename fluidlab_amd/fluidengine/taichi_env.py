"""TaichiEnv -- wires simulator, agent, bodies and loss together (fluidlab/fluidengine/taichi_env.py).

The name is kept so the reference's envs / optimiser import it unchanged; there is no Taichi here
(no ti.init, taichi_env.py:12): the simulator drives the MI355X HIP engine.  Renderers are outside this build
(#12-#13)."""
import numpy as np

from fluidlab_amd.configs.macros import DTYPE_NP
from fluidlab_amd.fluidengine import agents as _agents
from fluidlab_amd.fluidengine.bodies import Bodies
from fluidlab_amd.fluidengine.meshes import Statics
from fluidlab_amd.fluidengine.simulators import MPMSimulator, SmokeField
from fluidlab_amd.utils.config import CfgNode


class TaichiEnv:
    def __init__(self, dim=3, quality=1, particle_density=1e6, max_substeps_local=50, max_substeps_global=100000,
                 horizon=100, ckpt_dest='disk', gravity=(0.0, -10.0, 0.0), engine_lib=None, device=0):
        self.particle_density = particle_density
        self.dim = dim
        self.max_substeps_global = max_substeps_global
        self.horizon = horizon
        self.ckpt_dest = ckpt_dest
        self.t = 0
        self.simulator = MPMSimulator(dim=dim, quality=quality, horizon=horizon, max_substeps_local=max_substeps_local,
                                      max_substeps_global=max_substeps_global, gravity=gravity, ckpt_dest=ckpt_dest,
                                      engine_lib=engine_lib, device=device)
        self.max_substeps_local = self.simulator.max_substeps_local
        self.agent = None
        self.statics = Statics()
        self.particle_bodies = Bodies(dim=dim, particle_density=particle_density, elib=self.simulator.engine_library, device=device)
        self.renderer = None
        self.loss = None
        self.smoke_field = None
        print('===>  TaichiEnv created.')

    def setup_agent(self, agent_cfg):
        cls = getattr(_agents, agent_cfg.type)
        self.agent = cls(max_substeps_local=self.max_substeps_local, max_substeps_global=self.max_substeps_global,
                         max_action_steps_global=self.horizon, ckpt_dest=self.ckpt_dest, **agent_cfg.get('params', {}))
        for effector_cfg_dict in agent_cfg.effectors:
            effector_cfg = CfgNode(effector_cfg_dict)
            self.agent.add_effector(type=effector_cfg.type, params=dict(effector_cfg.params),
                                    mesh_cfg=effector_cfg.get('mesh', None), boundary_cfg=dict(effector_cfg.boundary))

    def setup_renderer(self, **kwargs):
        raise NotImplementedError('renderers are outside this build (SURVEY 2 #12-#13)')

    def setup_boundary(self, **kwargs):
        self.simulator.setup_boundary(**kwargs)

    def add_static(self, **kwargs):
        self.statics.add_static(**kwargs)

    def add_body(self, **kwargs):
        self.particle_bodies.add_body(**kwargs)

    def setup_smoke_field(self, **kwargs):
        self.smoke_field = SmokeField(dim=self.dim, ckpt_dest=self.ckpt_dest, **kwargs)          # taichi_env.py:95-100

    def setup_loss(self, loss_cls, **kwargs):
        self.loss = loss_cls(max_loss_steps=self.horizon, **kwargs)

    def build(self):
        self.particles = self.particle_bodies.get()
        self.n_particles = len(self.particles['x']) if self.particles is not None else 0
        self.has_particles = self.particles is not None
        self.simulator.build(self.agent, self.smoke_field, self.statics, self.particles)
        if self.smoke_field is not None:
            self.smoke_field.build(self.simulator, self.agent)                                   # taichi_env.py:125-126
        if self.agent is not None:
            self.agent.build(self.simulator)
        if self.loss is not None:
            self.loss.build(self.simulator)
        self.t = 0

    def reset_grad(self):
        self.simulator.reset_grad()
        if self.agent is not None:
            self.agent.reset_grad()
        if self.loss is not None:
            self.loss.reset_grad()

    def enable_grad(self):
        self.simulator.enable_grad()

    def disable_grad(self):
        self.simulator.disable_grad()

    @property
    def grad_enabled(self):
        return self.simulator.grad_enabled

    def render(self, mode='human', tgt_particles=None):
        raise AssertionError('No renderer available.')

    def get_state_RL(self):
        return self.simulator.get_state_RL()

    def step(self, action=None):
        if action is not None:
            assert self.agent is not None, 'Environment has no agent to execute action.'
            action = np.array(action).astype(DTYPE_NP)
        self.simulator.step(action=action)
        if self.loss:
            self.loss.step()
        self.t += 1

    def step_grad(self, action=None):
        if self.loss:
            self.loss.step_grad()
        if action is not None:
            assert self.agent is not None, 'Environment has no agent to execute action.'
            action = np.array(action).astype(DTYPE_NP)
        self.simulator.step_grad(action=action)

    def get_step_loss(self):
        assert self.loss is not None
        return self.loss.get_step_loss()

    def get_final_loss(self):
        assert self.loss is not None
        return self.loss.get_final_loss()

    def get_final_loss_grad(self):
        assert self.loss is not None
        self.loss.get_final_loss_grad()

    def get_state(self):
        return {'state': self.simulator.get_state(), 'grad_enabled': self.grad_enabled}

    def set_state(self, state, grad_enabled=False):
        self.t = 0
        self.simulator.cur_substep_global = 0
        self.simulator.set_state(0, state)
        if grad_enabled:
            self.enable_grad()
        else:
            self.disable_grad()
        if self.loss:
            self.loss.reset()

    def apply_agent_action_p(self, action_p):
        assert self.agent is not None, 'Environment has no agent to execute action.'
        self.agent.apply_action_p(action_p)

    def apply_agent_action_p_grad(self, action_p):
        assert self.agent is not None, 'Environment has no agent to execute action.'
        self.agent.apply_action_p_grad(action_p)
