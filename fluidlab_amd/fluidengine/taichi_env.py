"""TaichiEnv -- the object the envs and the optimiser talk to: it owns the simulator and hands agent, bodies, statics, smoke
field and loss to it (interface of fluidlab/fluidengine/taichi_env.py).

The name is kept so code written against the reference imports it unchanged; there is no Taichi here (no ti.init,
taichi_env.py:12): the simulator drives the MI355X HIP engine.  Renderers are outside this build (#12-#13)."""
import numpy as np

from fluidlab_amd.configs.macros import DTYPE_NP
from fluidlab_amd.fluidengine import agents as _agents
from fluidlab_amd.fluidengine.bodies import Bodies
from fluidlab_amd.fluidengine.meshes import Statics
from fluidlab_amd.fluidengine.simulators import MPMSimulator, SmokeField
from fluidlab_amd.utils.config import CfgNode

_NO_AGENT = 'Environment has no agent to execute action.'


class TaichiEnv:
    def __init__(self, dim=3, quality=1, particle_density=1e6, max_substeps_local=50, max_substeps_global=100000,
                 horizon=100, ckpt_dest='disk', gravity=(0.0, -10.0, 0.0), engine_lib=None, device=0, dt=None):
        self.dim, self.horizon, self.ckpt_dest = dim, horizon, ckpt_dest
        self.particle_density, self.max_substeps_global = particle_density, max_substeps_global
        self.simulator = MPMSimulator(dim=dim, quality=quality, horizon=horizon, max_substeps_local=max_substeps_local,
                                      max_substeps_global=max_substeps_global, gravity=gravity, ckpt_dest=ckpt_dest,
                                      engine_lib=engine_lib, device=device, dt=dt)
        self.max_substeps_local = self.simulator.max_substeps_local          # None (whole trajectory resident) is resolved there
        self.statics = Statics()
        self.particle_bodies = Bodies(dim=dim, particle_density=particle_density, elib=self.simulator.engine_library, device=device)
        self.agent = self.loss = self.smoke_field = self.renderer = None
        self.t = 0
        print('===>  TaichiEnv created.')

    # ---- scene description (called by FluidEnv.build_env before build())
    def setup_agent(self, agent_cfg):
        """agent_cfg: {type, params?, effectors: [{type, params, mesh?, boundary}]} (envs/configs/agent_*.yaml)"""
        make_agent = getattr(_agents, agent_cfg.type)
        self.agent = make_agent(max_substeps_local=self.max_substeps_local, max_substeps_global=self.max_substeps_global,
                                max_action_steps_global=self.horizon, ckpt_dest=self.ckpt_dest, **agent_cfg.get('params', {}))
        for entry in agent_cfg.effectors:
            eff = CfgNode(entry)
            self.agent.add_effector(type=eff.type, params=dict(eff.params), mesh_cfg=eff.get('mesh', None), boundary_cfg=dict(eff.boundary))

    def setup_boundary(self, **kwargs):
        self.simulator.setup_boundary(**kwargs)

    def setup_smoke_field(self, **kwargs):
        self.smoke_field = SmokeField(dim=self.dim, ckpt_dest=self.ckpt_dest, **kwargs)          # taichi_env.py:95-100

    def setup_loss(self, loss_cls, **kwargs):
        self.loss = loss_cls(max_loss_steps=self.horizon, **kwargs)

    def setup_renderer(self, **kwargs):
        raise NotImplementedError('renderers are outside this build (SURVEY 2 #12-#13)')

    def add_static(self, **kwargs):
        self.statics.add_static(**kwargs)

    def add_body(self, **kwargs):
        self.particle_bodies.add_body(**kwargs)

    def build(self):
        """simulator first (it creates the engine), then whoever registers with it: smoke field, agent, loss (taichi_env.py:108-134)"""
        self.particles = self.particle_bodies.get()
        self.has_particles = self.particles is not None
        self.n_particles = len(self.particles['x']) if self.has_particles else 0
        self.simulator.build(self.agent, self.smoke_field, self.statics, self.particles)
        for part, args in ((self.smoke_field, (self.simulator, self.agent)), (self.agent, (self.simulator,)), (self.loss, (self.simulator,))):
            if part is not None:
                part.build(*args)
        self.t = 0

    # ---- differentiation switches
    def reset_grad(self):
        for part in (self.simulator, self.agent, self.loss):
            if part is not None:
                part.reset_grad()

    def enable_grad(self):
        self.simulator.enable_grad()

    def disable_grad(self):
        self.simulator.disable_grad()

    @property
    def grad_enabled(self):
        return self.simulator.grad_enabled

    # ---- stepping
    def _as_action(self, action):
        if action is None:
            return None
        assert self.agent is not None, _NO_AGENT
        return np.array(action).astype(DTYPE_NP)

    def step(self, action=None):
        self.simulator.step(action=self._as_action(action))
        if self.loss:
            self.loss.step()
        self.t += 1

    def step_grad(self, action=None):
        if self.loss:                                   # the loss of a step is differentiated before the step itself
            self.loss.step_grad()
        self.simulator.step_grad(action=self._as_action(action))

    def apply_agent_action_p(self, action_p):
        assert self.agent is not None, _NO_AGENT
        self.agent.apply_action_p(action_p)

    def apply_agent_action_p_grad(self, action_p):
        assert self.agent is not None, _NO_AGENT
        self.agent.apply_action_p_grad(action_p)

    # ---- loss and state access
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
        """restart from `state` at substep 0 (taichi_env.py:203-215)"""
        self.t = 0
        self.simulator.cur_substep_global = 0
        self.simulator.set_state(0, state)
        (self.enable_grad if grad_enabled else self.disable_grad)()
        if self.loss:
            self.loss.reset()

    def get_state_RL(self):
        return self.simulator.get_state_RL()

    def render(self, mode='human', tgt_particles=None):
        raise AssertionError('No renderer available.')
