"""Effector -- action -> effector pose, and the action-gradient read-out (fluidlab/fluidengine/effectors/effector.py).

The reference's 12 one-thread Taichi kernels (set_velocity, move_kernel, apply_action_p_kernel, their .grad's,
copy/ckpt helpers) live inside the engine; move_kernel and its adjoint are folded into the p2g / p2g_grad
launches.  This class keeps the reference's method surface and forwards to the engine by effector index."""
import numpy as np

from fluidlab_amd import _capi
from fluidlab_amd.fluidengine.boundaries import create_boundary
from fluidlab_amd.utils.geom import euler_to_quat_wxyz
from fluidlab_amd.configs.macros import DTYPE_NP
from fluidlab_amd.utils.misc import eval_str


class _Field1:
    def __init__(self, a):
        self._a = a

    def to_numpy(self):
        return self._a


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

    @property
    def latest_pos(self):
        """effector.py:146-152: `move(f)` stores pos[f] -- the pose at the START of the last substep that moved the effector -- in a
        1-element field for the renderer; the Gathering / Mixing policies steer by it (policies.py:240, 327).  Served from the
        engine: the simulator remembers the local frame of the last move (None before the first one: the current pose, which is
        what the field holds after apply_action_p in this package).  A frame that old survives until the window has gone round
        once more; policies read the value right after the step that moved."""
        f = getattr(self.sim, 'last_move_f', None)
        if f is None:
            f = self.sim.cur_substep_local
        pos = self.engine.eff_get_state(self.index, f)[:3]
        return _Field1(np.asarray(pos, np.float32)[None, :])

    # ---- state (effector.py:185-208)
    def get_state(self, f):
        return self.engine.eff_get_state(self.index, f)[:7]

    def set_state(self, f, state):
        full = self.engine.eff_get_state(self.index, f)
        full[:len(state)] = state
        self.engine.eff_set_state(self.index, f, full)

    # ---- checkpoints (effector.py:84-140): frame-0 pose + velocities
    # checkpoint payload of frame 0, the reference's wire format (effector.py:103-140; Injector adds 'act_id',
    # injector.py:131-171), so chunk files written by either implementation load in the other
    has_act_id = False

    def get_ckpt(self, ckpt_name=None):
        st = self.engine.eff_get_state(self.index, 0)
        v, w = self.engine.eff_get_vw(self.index, 0)
        dt = self.engine.dtype          # fp32 in the product (DTYPE_NP, as the reference); the fp64 oracle keeps its precision
        ckpt = {'pos': st[:3].astype(dt), 'quat': st[3:7].astype(dt), 'v': np.asarray(v, dt), 'w': np.asarray(w, dt)}
        if self.has_act_id:
            ckpt['act_id'] = np.int32(round(float(st[7])))
        return ckpt

    def set_ckpt(self, ckpt=None, ckpt_name=None):
        act_id = float(ckpt['act_id']) if 'act_id' in ckpt else float(self.engine.eff_get_state(self.index, 0)[7])
        st = np.concatenate([np.asarray(ckpt['pos'], np.float64).reshape(3), np.asarray(ckpt['quat'], np.float64).reshape(4), [act_id]])
        self.engine.eff_set_state(self.index, 0, st)
        self.engine.eff_set_vw(self.index, 0, np.asarray(ckpt['v']).reshape(3), np.asarray(ckpt['w']).reshape(3))

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
