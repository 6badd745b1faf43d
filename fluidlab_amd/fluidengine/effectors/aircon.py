"""AirCon -- the blowing effector of Circulation-v0 (fluidlab/fluidengine/effectors/aircon.py): an 8-dof action, the two
extra entries being the jet strength s and its radius r (aircon.py:205-213), which the smoke field's impulse reads
(smoke_field.py:213-218).  State is 9-dimensional (pos, quat, s, r; aircon.py:178-203)."""
import numpy as np

from fluidlab_amd import _capi
from fluidlab_amd.utils.misc import eval_str
from .effector import Effector


class AirCon(Effector):
    state_dim = 9
    abi_type = _capi.FE_EFF_AIRCON

    def __init__(self, inject_v=(-0.3, 0.0, 1.0), **kwargs):
        super().__init__(**kwargs)
        self.has_dynamics = False
        self.mesh = None
        self.inject_v = np.asarray(eval_str(inject_v), np.float64)

    def setup_mesh(self, **kwargs):
        self.mesh = dict(kwargs, has_dynamics=False)             # visual only (aircon.py:28-33)

    def _abi_desc(self, elib):
        d = super()._abi_desc(elib)
        d['inject_v'] = tuple(float(t) for t in self.inject_v)
        return d

    def get_state(self, f):
        st = self.engine.eff_get_state(self.index, f)
        s, r = self.engine.eff_get_sr(self.index, f)
        return np.concatenate([st[:7], [s, r]])

    def set_state(self, f, state):
        full = self.get_state(f)
        full[:len(state)] = state
        st = self.engine.eff_get_state(self.index, f)
        st[:7] = full[:7]
        self.engine.eff_set_state(self.index, f, st)
        self.engine.eff_set_sr(self.index, f, full[7], full[8])

    @property
    def init_state(self):
        return np.append(self.init_pos, self.init_rot)

    def get_ckpt(self, ckpt_name=None):
        ckpt = super().get_ckpt(ckpt_name)
        s, r = self.engine.eff_get_sr(self.index, 0)
        dt = self.engine.dtype
        ckpt['s'], ckpt['r'] = np.asarray(s, dt), np.asarray(r, dt)                  # aircon.py:104-113
        return ckpt

    def set_ckpt(self, ckpt=None, ckpt_name=None):
        super().set_ckpt(ckpt, ckpt_name)
        self.engine.eff_set_sr(self.index, 0, float(ckpt['s']), float(ckpt['r']))
