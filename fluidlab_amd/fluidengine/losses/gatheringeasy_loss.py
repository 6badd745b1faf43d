"""GatheringEasyLoss (fluidlab/fluidengine/losses/gatheringeasy_loss.py): sum over the matching material's used particles of
|x_0 - 0.8| -- push the floating bodies towards x = 0.8 (gatheringeasy_loss.py:75-79).  A HostLoss: evaluated next to the
frames (torch on the GPU for the HIP engine, numpy against the oracle)."""
import numpy as np

from .host_loss import HostLoss


class GatheringEasyLoss(HostLoss):
    goal_x = 0.8

    def __init__(self, type, matching_mat, **kwargs):
        super().__init__(**kwargs)
        self.matching_mat = matching_mat
        if type == 'diff':
            self.plateau_count_limit = 10
            self.temporal_expand_speed = 120
            self.temporal_init_range_end = 120
            self.temporal_range_type = 'expand'
        elif type == 'default':
            self.temporal_range_type = 'all'
        else:
            assert False

    def build(self, sim):
        self.dist_weight = self.weights['dist']
        super().build(sim)

    def step_value(self, s, f, x, used, want_grad):
        xp = self.xp
        m = used & (self.particle_mat == self.matching_mat)
        d = x[m, 0] - self.goal_x
        g = None
        if want_grad:
            g = xp.zeros_like(x)
            g[m, 0] = xp.sign(d) * self.dist_weight
        return xp.to_float(xp.abs(d).sum()) * self.dist_weight, g

    def final_loss_info(self):
        return {'reward': float(np.sum((150 - self._step_loss) * 0.01))}

    def get_step_loss(self):
        cur = self.cur_step_loss()
        return {'reward': 0.01 * (150 - cur), 'loss': 0.01 * cur}                       # gatheringeasy_loss.py:113-121
