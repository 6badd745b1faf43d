"""ShapeMatchingLoss -- squared distance of the matching material's particles to a recorded target
(fluidlab/fluidengine/losses/shapematching_loss.py), with the temporal-range curriculum."""
import pickle as pkl

import numpy as np

from .loss import Loss


class ShapeMatchingLoss(Loss):
    def __init__(self, matching_mat, temporal_range_type='expand', temporal_init_range_end=50, plateau_count_limit=5,
                 temporal_expand_speed=50, plateau_thresh=(0.01, 0.5), **kwargs):
        super().__init__(**kwargs)
        self.matching_mat = matching_mat
        self.temporal_range_type = temporal_range_type
        self.temporal_init_range_end = temporal_init_range_end
        self.plateau_count_limit = plateau_count_limit
        self.temporal_expand_speed = temporal_expand_speed
        self.plateau_thresh = plateau_thresh
        self.target = None

    def build(self, sim):
        self.chamfer_weight = self.weights['chamfer']
        if self.temporal_range_type == 'last':
            self.temporal_range = [self.max_loss_steps - 1, self.max_loss_steps]
        elif self.temporal_range_type == 'all':
            self.temporal_range = [0, self.max_loss_steps]
        elif self.temporal_range_type == 'expand':
            self.temporal_range = [0, self.temporal_init_range_end]
            self.best_loss = self.inf
            self.plateau_count = 0
        super().build(sim)

    def load_target(self, path):
        """The whole target trajectory is uploaded once and stays resident in HBM; the reference re-uploads one
        step per loss evaluation, forward and backward (shapematching_loss.py:73,77)."""
        target = path if isinstance(path, dict) else pkl.load(open(path, 'rb'))
        self.set_target(target)
        if not isinstance(path, dict):
            print(f'===>  Target loaded from {path}.')

    def set_target(self, target):
        assert self.max_loss_steps == len(target['x'])
        assert self.n_particles == len(target['x'][0])
        self.target = target
        for s in range(self.max_loss_steps):
            self.engine.loss_set_target(s, np.asarray(target['x'][s], dtype=self.engine.dtype))

    def compute_step_loss(self, s, f):
        self.engine.loss_step(s, f, self.matching_mat, self.chamfer_weight)        # chamfer kernel + sum_up (80-88)

    def compute_step_loss_grad(self, s, f):
        self.engine.loss_step_grad(s, f, self.matching_mat, self.chamfer_weight, float(self.step_loss_grad[s]))

    def _total_loss(self):
        sl = self.step_loss
        total = np.float32(0.0)
        for s in range(self.temporal_range[0], self.temporal_range[1]):            # compute_total_loss_kernel (90-93)
            total += sl[s]
        return float(total)

    def get_final_loss(self):
        self.total_loss = self._total_loss()
        self.expand_temporal_range()
        return {'loss': self.total_loss, 'last_step_loss': float(self.step_loss[self.max_loss_steps - 1]),
                'temporal_range': self.temporal_range[1]}

    def get_final_loss_grad(self):
        # compute_total_loss_kernel.grad: step_loss.grad[s] += total_loss.grad for s in the temporal range
        self.step_loss_grad[self.temporal_range[0]:self.temporal_range[1]] += self.total_loss_grad

    def expand_temporal_range(self):
        """shapematching_loss.py:110-128"""
        if self.temporal_range_type != 'expand':
            return
        loss_improved = self.best_loss - self.total_loss
        loss_improved_rate = loss_improved / self.best_loss if self.best_loss != 0 else 0.0     # (a loss of exactly 0 only in tests)
        if loss_improved_rate < self.plateau_thresh[0] or loss_improved < self.plateau_thresh[1]:
            self.plateau_count += 1
            print('Plateaued!!!', self.plateau_count)
        else:
            self.plateau_count = 0
        if self.best_loss > self.total_loss:
            self.best_loss = self.total_loss
        if self.plateau_count >= self.plateau_count_limit:
            self.plateau_count = 0
            self.best_loss = self.inf
            self.temporal_range[1] = min(self.max_loss_steps, self.temporal_range[1] + self.temporal_expand_speed)
            print(f'temporal range expanded to {self.temporal_range}')
