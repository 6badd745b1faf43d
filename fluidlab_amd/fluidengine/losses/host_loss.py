"""HostLoss -- shared plumbing of the task losses whose per-step term is a small reduction over particle positions
(gatheringeasy / gatheringO / pouring / transporting / mixing _loss.py in fluidlab/fluidengine/losses).  The engine owns the
frames; a step downloads x/used of one frame (`fe_get_frame`), the subclass returns the step's value and d value / d x, and
the backward step uploads the adjoint (`fe_add_grad`).  Temporal-range handling ('last' / 'all' / 'expand' with the plateau
rule) is the block every one of those reference files repeats."""
import numpy as np

from .loss import Loss


def pairwise_l1(a, b=None):
    """sum_{i,j} |a_i - b_j|_1 and its gradients without forming the pairs: per dimension the sum separates after a sort
    (value by prefix sums, d/d a_i = #{b < a_i} - #{b > a_i}).  b=None: all ordered pairs (i, j) of `a` itself
    (mixing_loss.py:74-75), where every unordered pair appears twice.  float64 in and out."""
    a = np.asarray(a, np.float64)
    self_pairs = b is None
    b = a if self_pairs else np.asarray(b, np.float64)
    total, ga, gb = 0.0, np.zeros_like(a), np.zeros_like(b)
    if len(a) == 0 or len(b) == 0:
        return total, ga, (None if self_pairs else gb)
    for d in range(a.shape[1]):
        bs = np.sort(b[:, d])
        csum = np.concatenate([[0.0], np.cumsum(bs)])
        lo, hi = np.searchsorted(bs, a[:, d], 'left'), np.searchsorted(bs, a[:, d], 'right')
        # b < a_i : lo of them (sum csum[lo]); b > a_i : len - hi of them
        total += float((a[:, d] * lo - csum[lo]).sum() + ((csum[-1] - csum[hi]) - a[:, d] * (len(bs) - hi)).sum())
        ga[:, d] = lo - (len(bs) - hi)
        if not self_pairs:
            as_ = np.sort(a[:, d])
            lo_b, hi_b = np.searchsorted(as_, b[:, d], 'left'), np.searchsorted(as_, b[:, d], 'right')
            gb[:, d] = lo_b - (len(as_) - hi_b)
    if self_pairs:
        return total, 2.0 * ga, None          # a_i appears as first and as second argument
    return total, ga, gb


class HostLoss(Loss):
    temporal_range_type = 'all'
    plateau_count_limit = 10
    temporal_expand_speed = 0
    temporal_init_range_end = 0
    plateau_thresh = (1e-6, 0.1)

    def build(self, sim):
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
        self._mat = None

    @property
    def step_loss(self):
        return self._step_loss

    @property
    def particle_mat(self):
        if self._mat is None:
            self._mat = self.sim.particles_i.mat.to_numpy()
        return self._mat

    def clear_loss(self):
        super().clear_loss()
        if hasattr(self, '_step_loss'):
            self._step_loss[:] = 0
            self.total_loss = 0.0

    def frame(self, f):
        x = np.zeros((self.n_particles, 3), self.engine.dtype); used = np.zeros((self.n_particles,), np.int32)
        self.engine.get_frame(f, x=x, used=used)
        return x, used > 0

    # subclasses: value of step s at frame f and (optionally) its gradient d value / d x as an [N, 3] array or None
    def step_value(self, s, f, x, used, want_grad):
        raise NotImplementedError

    def compute_step_loss(self, s, f):
        x, used = self.frame(f)
        value, _ = self.step_value(s, f, x, used, False)
        self._step_loss[s] += float(value)

    def compute_step_loss_grad(self, s, f):
        if not (self.temporal_range[0] <= s < self.temporal_range[1]):
            return
        x, used = self.frame(f)
        _, gx = self.step_value(s, f, x, used, True)
        if gx is not None:
            self.engine.add_grad(f, gx=(gx * self.total_loss_grad).astype(self.engine.dtype))

    def final_loss_info(self):
        return {}

    def get_final_loss(self):
        self.total_loss = float(self._step_loss[self.temporal_range[0]:self.temporal_range[1]].sum())
        self.expand_temporal_range()
        info = {'loss': self.total_loss, 'last_step_loss': float(self._step_loss[self.max_loss_steps - 1]),
                'temporal_range': self.temporal_range[1]}
        info.update(self.final_loss_info())
        return info

    def get_final_loss_grad(self):
        pass                                              # applied per step in compute_step_loss_grad

    def expand_temporal_range(self):
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

    def cur_step_loss(self):
        return float(self._step_loss[self.sim.cur_step_global - 1])
