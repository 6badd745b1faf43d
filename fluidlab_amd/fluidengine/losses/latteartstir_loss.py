"""LatteArtStirLoss (fluidlab/fluidengine/losses/latteartstir_loss.py): the shape-matching distance over *all* used particles
drives the optimisation (latteartstir_loss.py:62-68); the MILK_VIS-only part is reported as `loss_milk`."""
import numpy as np

from fluidlab_amd.configs.macros import MILK_VIS
from .shapematching_loss import ShapeMatchingLoss


class LatteArtStirLoss(ShapeMatchingLoss):
    def __init__(self, type, **kwargs):
        if type == 'diff':
            super().__init__(matching_mat=MILK_VIS, temporal_init_range_end=50, temporal_range_type='expand', plateau_count_limit=5,
                             temporal_expand_speed=10, plateau_thresh=[0.01, 0.1], **kwargs)
        elif type == 'default':
            super().__init__(matching_mat=MILK_VIS, temporal_range_type='all', **kwargs)
        else:
            assert False

    def build(self, sim):
        super().build(sim)
        self.step_loss_milk = np.zeros((self.max_loss_steps,), np.float64)
        self._milk = None

    def clear_loss(self):
        super().clear_loss()
        if hasattr(self, 'step_loss_milk'):
            self.step_loss_milk[:] = 0

    def compute_step_loss(self, s, f):
        self.engine.loss_step(s, f, -1, self.chamfer_weight)                       # every used particle
        if self.target is not None:                                                # the milk-only figure: host side, reporting only
            if self._milk is None:
                self._milk = self.sim.particles_i.mat.to_numpy() == MILK_VIS
            x = np.zeros((self.n_particles, 3), self.engine.dtype); used = np.zeros((self.n_particles,), np.int32)
            self.engine.get_frame(f, x=x, used=used)
            m = self._milk & (used > 0)
            d = x[m].astype(np.float64) - np.asarray(self.target['x'][s])[m]
            self.step_loss_milk[s] += float((d * d).sum()) * self.chamfer_weight

    def compute_step_loss_grad(self, s, f):
        self.engine.loss_step_grad(s, f, -1, self.chamfer_weight, float(self.step_loss_grad[s]))

    def get_final_loss(self):
        info = super().get_final_loss()
        info['loss_milk'] = float(self.step_loss_milk[self.temporal_range[0]:self.temporal_range[1]].sum())
        info['reward'] = float(np.sum((1000 - self.step_loss) * 0.002))
        return info

    def get_step_loss(self):
        cur_step_loss = float(self.step_loss[self.sim.cur_step_global - 1])
        return {'reward': 0.002 * (1000 - cur_step_loss), 'loss': 0.002 * cur_step_loss}
