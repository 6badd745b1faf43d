"""CirculationLoss -- room-temperature objective of Circulation-v0 (fluidlab/fluidengine/losses/circulation_loss.py):
|T - 1| at five detector cells, |T - target_temp| at ten others, every step (circulation_loss.py:97-109).  Fifteen cells
per step: the reduction and its adjoint run on the host; the temperature field and its adjoint stay in the engine."""
import numpy as np

from .loss import Loss


class CirculationLoss(Loss):
    def __init__(self, type, detectors=None, **kwargs):
        super().__init__(**kwargs)
        self.plateau_count_limit = 10
        self.temporal_expand_speed = 0
        self.temporal_init_range_end = 0
        self.temporal_range_type = 'all'
        self.plateau_thresh = [1e-6, 0.1]
        self._detectors = detectors

    def build(self, sim):
        self.temp_weight = self.weights['temp']
        self.temporal_range = [0, self.max_loss_steps]                                   # 'all' (circulation_loss.py:33-41)
        self.target_temp = 0.0
        self.detector_h = 64
        h = self.detector_h
        default = [[25, h, 85], [35, h, 85], [15, h, 85], [25, h, 75], [25, h, 95],      # circulation_loss.py:46-65
                   [25, h, 42], [35, h, 42], [15, h, 42], [25, h, 32], [25, h, 52],
                   [107, h, 65], [115, h, 65], [99, h, 65], [107, h, 45], [107, h, 85]]
        self.detector_array = np.asarray(default if self._detectors is None else self._detectors, np.int32)
        self.detector_array_N = len(self.detector_array)
        self.smoke_field = sim.smoke_field                                               # loss.py:41-42
        self.temp_loss = np.zeros((self.max_loss_steps,), np.float64)
        self._step_loss = np.zeros((self.max_loss_steps,), np.float64)
        self.total_loss = 0.0
        super().build(sim)

    @property
    def step_loss(self):
        return self._step_loss

    def clear_loss(self):
        super().clear_loss()
        if hasattr(self, '_step_loss'):
            self._step_loss[:] = 0
            self.total_loss = 0.0

    def clear_losses(self):
        if hasattr(self, 'temp_loss'):
            self.temp_loss[:] = 0

    def _targets(self):
        t = np.full(self.detector_array_N, self.target_temp)
        t[:5] = 1.0
        return t

    def step(self):
        """circulation_loss.py:77-79: loss of step cur_step_global - 1, read at local step frame cur_step_local"""
        s_global, s_local = self.sim.cur_step_global - 1, self.sim.cur_step_local
        q = self.smoke_field.q_at(s_local, self.detector_array)
        self.temp_loss[s_global] += np.abs(q - self._targets()).sum()
        self._step_loss[s_global] += self.temp_loss[s_global] * self.temp_weight

    def step_grad(self):
        s_global, s_local = self.sim.cur_step_global - 1, self.sim.cur_step_local
        if not (self.temporal_range[0] <= s_global < self.temporal_range[1]):
            return
        q = self.smoke_field.q_at(s_local, self.detector_array)
        g = np.sign(q - self._targets()) * self.temp_weight * self.total_loss_grad       # d|.| = sign (0 at the kink, as Taichi)
        self.smoke_field.add_q_grad_at(s_local, self.detector_array, g)

    def get_final_loss(self):
        self.total_loss = float(self._step_loss[self.temporal_range[0]:self.temporal_range[1]].sum())
        # temporal_range_type is 'all' (circulation_loss.py:25): expand_temporal_range has nothing to expand
        return {'loss': self.total_loss, 'last_step_loss': float(self._step_loss[self.max_loss_steps - 1]),
                'temporal_range': self.temporal_range[1]}

    def get_final_loss_grad(self):
        pass                                            # step_loss.grad[s] = total_loss.grad for s in the range: applied in step_grad

    def get_step_loss(self):
        cur = float(self._step_loss[self.sim.cur_step_global - 1])
        return {'reward': 1.0 * (11 - cur), 'loss': 1.0 * cur}
