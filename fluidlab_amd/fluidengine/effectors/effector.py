"""Effector -- action -> effector pose, and the action-gradient read-out (fluidlab/fluidengine/effectors/effector.py).

The reference's 12 one-thread Taichi kernels (set_velocity, move_kernel, apply_action_p_kernel, their .grad's,
copy/ckpt helpers) live inside the engine; move_kernel and its adjoint are folded into the p2g / p2g_grad
launches.  This class keeps the reference's method surface and forwards to the engine by effector index."""
import numpy as np

from fluidlab_amd import _capi
from fluidlab_amd.fluidengine.boundaries import create_boundary
from fluidlab_amd.utils.geom import euler_to_quat_wxyz
from fluidlab_amd.utils.misc import eval_str


class Effector:
    state_dim = 7
    abi_type = _capi.FE_EFF_PLAIN

    def __init__(self, max_substeps_local, max_substeps_global, max_action_steps_global, ckpt_dest, dim=3, action_dim=3,
                 action_scale_p=(1.0, 1.0, 1.0), action_scale_v=(1.0, 1.0, 1.0), init_pos=(0.5, 0.5, 0.5),
                 init_euler=(0.0, 0.0, 0.0)):
        self.dim = dim
        self.max_substeps_local = max_substeps_local
        self.max_substeps_global = max_substeps_global
        self.max_action_steps_global = max_action_steps_global
        self.ckpt_dest = ckpt_dest
        self.action_dim = action_dim
        self.action_scale_v = tuple(eval_str(action_scale_v))
        self.action_scale_p = tuple(eval_str(action_scale_p))
        self.init_pos = np.array(eval_str(init_pos))
        self.init_rot = euler_to_quat_wxyz(eval_str(init_euler))            # effector.py:45
        self.boundary = None
        self.mesh = None
        self.engine = None
        self.index = None

    def setup_boundary(self, **kwargs):
        self.boundary = create_boundary(**kwargs)

    def setup_mesh(self, **kwargs):
        self.mesh = kwargs           # visual only (effector meshes are renderer-side; missing assets)

    # ---- registration with the engine
    def _abi_desc(self, elib):
        return dict(type=self.abi_type, action_dim=self.action_dim, action_scale_v=self.action_scale_v,
                    action_scale_p=self.action_scale_p, boundary=self.boundary.to_abi(elib))

    def build(self, engine):
        self.engine = engine
        self.index = engine.add_effector(**self._abi_desc(engine.elib))
        self.set_state(0, self.init_state)                                  # effector.py:215-216

    @property
    def init_state(self):
        return np.append(self.init_pos, self.init_rot)

    # ---- state (effector.py:185-208)
    def get_state(self, f):
        return self.engine.eff_get_state(self.index, f)[:7]

    def set_state(self, f, state):
        full = self.engine.eff_get_state(self.index, f)
        full[:len(state)] = state
        self.engine.eff_set_state(self.index, f, full)

    # ---- checkpoints (effector.py:84-140): frame-0 pose + velocities
    def get_ckpt(self):
        v, w = self.engine.eff_get_vw(self.index, 0)
        return {'state': self.engine.eff_get_state(self.index, 0), 'v': v, 'w': w}

    def set_ckpt(self, ckpt):
        self.engine.eff_set_state(self.index, 0, ckpt['state'])
        self.engine.eff_set_vw(self.index, 0, ckpt['v'], ckpt['w'])

    # ---- actions (effector.py:218-283)
    def set_action(self, s, s_global, n_substeps, action):
        assert s_global <= self.max_action_steps_global
        assert s * n_substeps <= self.max_substeps_local
        if self.action_dim > 0:
            self.engine.eff_set_action(self.index, s, s_global, n_substeps, action)

    def set_action_grad(self, s, s_global, n_substeps, action):
        assert s_global <= self.max_action_steps_global
        assert s * n_substeps <= self.max_substeps_local
        if self.action_dim > 0:
            self.engine.eff_set_action_grad(self.index, s, s_global, n_substeps)

    def apply_action_p(self, action_p):
        if self.action_dim > 0:
            self.engine.eff_apply_action_p(self.index, np.asarray(action_p, dtype=self.engine.dtype))

    def apply_action_p_grad(self, action_p):
        if self.action_dim > 0:
            self.engine.eff_apply_action_p_grad(self.index)

    def get_action_grad(self, s, n):
        if self.action_dim > 0:
            return self.engine.eff_get_action_grad(self.index, s, n, self.action_dim)
        return None

    @property
    def latest_pos(self):
        raise NotImplementedError('rendering helper (effector.py:151-152): no renderer in this package')
