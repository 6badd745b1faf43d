"""GatheringEasyLoss (fluidlab/fluidengine/losses/gatheringeasy_loss.py): sum over the matching material's used particles of
|x_0 - 0.8| -- push the floating bodies towards x = 0.8 (gatheringeasy_loss.py:75-79).  The reduction is an N-wide pass per
step over data the engine owns; it runs on the host here (one frame download per step, one adjoint upload per backward step)."""
import numpy as np

from .loss import Loss


class GatheringEasyLoss(Loss):
    goal_x = 0.8

    def __init__(self, type, matching_mat, **kwargs):
        super().__init__(**kwargs)
        self.matching_mat = matching_mat
        if type == 'diff':
            self.plateau_count_limit = 10
            self.temporal_expand_speed = 120
            self.temporal_init_range_end = 120
            self.temporal_range_type = 'expand'
            self.plateau_thresh = [1e-6, 0.1]
        elif type == 'default':
            self.temporal_range_type = 'all'
        else:
            assert False

    def build(self, sim):
        self.dist_weight = self.weights['dist']
        if self.temporal_range_type == 'last':
            self.temporal_range = [self.max_loss_steps - 1, self.max_loss_steps]
        elif self.temporal_range_type == 'all':
            self.temporal_range = [0, self.max_loss_steps]
        elif self.temporal_range_type == 'expand':
            self.temporal_range = [0, min(self.temporal_init_range_end, self.max_loss_steps)]
            self.best_loss = self.inf
            self.plateau_count = 0
        self._step_loss = np.zeros((self.max_loss_steps,), np.float64)
        self.total_loss = 0.0
        super().build(sim)
        self._sel = None

    @property
    def step_loss(self):
        return self._step_loss

    def clear_loss(self):
        super().clear_loss()
        if hasattr(self, '_step_loss'):
            self._step_loss[:] = 0
            self.total_loss = 0.0

    def _frame(self, f):
        if self._sel is None:
            self._sel = self.sim.particles_i.mat.to_numpy() == self.matching_mat
        x = np.zeros((self.n_particles, 3), self.engine.dtype); used = np.zeros((self.n_particles,), np.int32)
        self.engine.get_frame(f, x=x, used=used)
        return x, self._sel & (used > 0)

    def compute_step_loss(self, s, f):
        x, m = self._frame(f)
        self._step_loss[s] += float(np.abs(x[m, 0].astype(np.float64) - self.goal_x).sum()) * self.dist_weight

    def compute_step_loss_grad(self, s, f):
        if not (self.temporal_range[0] <= s < self.temporal_range[1]):
            return
        x, m = self._frame(f)
        gx = np.zeros((self.n_particles, 3), self.engine.dtype)
        gx[m, 0] = np.sign(x[m, 0] - self.goal_x) * self.dist_weight * self.total_loss_grad
        self.engine.add_grad(f, gx=gx)

    def get_final_loss(self):
        self.total_loss = float(self._step_loss[self.temporal_range[0]:self.temporal_range[1]].sum())
        self.expand_temporal_range()
        return {'loss': self.total_loss, 'last_step_loss': float(self._step_loss[self.max_loss_steps - 1]),
                'temporal_range': self.temporal_range[1], 'reward': float(np.sum((150 - self._step_loss) * 0.01))}

    def get_final_loss_grad(self):
        pass                                              # applied per step in compute_step_loss_grad

    def expand_temporal_range(self):
        """gatheringeasy_loss.py:93-110"""
        if self.temporal_range_type != 'expand':
            return
        loss_improved = self.best_loss - self.total_loss
        loss_improved_rate = loss_improved / self.best_loss if self.best_loss != 0 else 0.0
        if loss_improved_rate < self.plateau_thresh[0] or loss_improved < self.plateau_thresh[1]:
            self.plateau_count += 1
        else:
            self.plateau_count = 0
        if self.best_loss > self.total_loss:
            self.best_loss = self.total_loss
        if self.plateau_count >= self.plateau_count_limit:
            self.plateau_count = 0
            self.best_loss = self.inf
            self.temporal_range[1] = min(self.max_loss_steps, self.temporal_range[1] + self.temporal_expand_speed)

    def get_step_loss(self):
        cur = float(self._step_loss[self.sim.cur_step_global - 1])
        return {'reward': 0.01 * (150 - cur), 'loss': 0.01 * cur}                       # gatheringeasy_loss.py:113-121
