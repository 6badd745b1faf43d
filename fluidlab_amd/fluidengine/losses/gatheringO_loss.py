"""GatheringOLoss (fluidlab/fluidengine/losses/gatheringO_loss.py): squared distance, in the xz plane, of every used particle
of the matching material to the goal (0.88, 0.78) (:75-79)."""
from .host_loss import HostLoss


class GatheringOLoss(HostLoss):
    goal = (0.88, 0.78)

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
        dx, dz = x[m, 0] - self.goal[0], x[m, 2] - self.goal[1]
        g = None
        if want_grad:
            g = xp.zeros_like(x)
            g[m, 0] = 2 * dx * self.dist_weight
            g[m, 2] = 2 * dz * self.dist_weight
        return xp.to_float((dx * dx + dz * dz).sum()) * self.dist_weight, g

    def get_step_loss(self):
        cur = self.cur_step_loss()
        return {'reward': 0.01 * (65 - cur), 'loss': 0.01 * cur}                  # :125-132
