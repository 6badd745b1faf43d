"""PouringLoss (fluidlab/fluidengine/losses/pouring_loss.py): keep the MILK where it started (L1 to the initial positions,
:131-135), pull the WATER towards the floor y = 0.05 ('default' only: dist_scale 0.2, :30-33) and, in the 'diff' variant, on
the last step attract the 100 WATER particles nearest to the lowest one towards it (:101-118, 137-146)."""
import numpy as np

from fluidlab_amd.configs.macros import MILK, WATER
from .host_loss import HostLoss


class PouringLoss(HostLoss):
    temporal_range_type = 'all'

    def __init__(self, type, **kwargs):
        super().__init__(**kwargs)
        self.type = type
        assert type in ('diff', 'default')
        self.dist_scale = 0.0 if type == 'diff' else 0.2

    def build(self, sim):
        self.dist_weight = self.weights['dist']
        self.attraction_weight = self.weights['attraction']
        self.init_particle_pos = None
        super().build(sim)

    def _attraction(self, x, used, want_grad):
        """find_best_particle + compute_attraction_loss_kernel + scale_attraction_loss_kernel (:101-146)"""
        water = used & (self.particle_mat == WATER)
        if not water.any():
            return 0.0, None
        xd = x.astype(np.float64)
        loss = np.abs(xd[:, 1] - 0.05); loss[~water] = 1000
        best = int(np.argmin(loss))
        dist = np.linalg.norm(xd - xd[best], axis=1); dist[~water] = 1000
        score = np.argsort(np.argsort(dist))
        near = water & (score < 100)
        scale = float((xd[water, 1] > 0.55).sum()) / 12500
        value = float(np.abs(xd[near] - xd[best]).sum()) * 5000 * scale
        if not want_grad:
            return value, None
        g = np.zeros_like(xd)
        g[near] = np.sign(xd[near] - xd[best]) * 5000 * scale          # best_particle_pos is a constant copy (:108)
        return value, g

    def step_value(self, s, f, x, used, want_grad):
        if s == 0 and not want_grad:
            self.init_particle_pos = x.astype(np.float64).copy()                  # get_init_particles, :97-99
        xd = x.astype(np.float64)
        water = used & (self.particle_mat == WATER)
        milk = used & (self.particle_mat == MILK)
        value = float(np.abs(xd[water, 1] - 0.05).sum()) * self.dist_scale + float(np.abs(xd[milk] - self.init_particle_pos[milk]).sum())
        value *= self.dist_weight
        g = None
        if want_grad:
            g = np.zeros_like(xd)
            g[water, 1] = np.sign(xd[water, 1] - 0.05) * self.dist_scale * self.dist_weight
            g[milk] = np.sign(xd[milk] - self.init_particle_pos[milk]) * self.dist_weight
        if self.type == 'diff' and s == self.max_loss_steps - 1:
            av, ag = self._attraction(x, used, want_grad)
            value += av
            if ag is not None:
                g += ag
        # sum_up_loss_kernel adds the attraction WEIGHT as a constant (`+ self.attraction_weight`, :150): kept, it shifts the
        # reported loss by one per step and has no gradient
        value += self.attraction_weight
        return value, g

    def get_step_loss(self):
        cur = self.cur_step_loss()
        return {'reward': 0.001 * (5000 - cur), 'loss': 0.001 * cur}             # :196-202
