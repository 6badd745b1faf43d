"""NumPy optimisers for the action sequence (fluidlab/optimizer/optim.py).  fp64 state, identical update
rule, so replicated ranks that apply the same all-reduced gradient stay bit-identical."""
import numpy as np


class Optimizer:
    def __init__(self, parameters_shape, cfg):
        self.cfg = cfg
        self.lr = self.init_lr = cfg.lr
        self.parameters_shape = parameters_shape
        self.initialize()

    def initialize(self):
        raise NotImplementedError

    def step(self, parameters, grads):
        return self._step(parameters, grads)


class Adam(Optimizer):
    def initialize(self):
        self.momentum_buffer = np.zeros(self.parameters_shape, dtype=np.float64)
        self.v_buffer = np.zeros(self.parameters_shape, dtype=np.float64)
        self.iter = 0

    def _step(self, parameters, grads):
        b1, b2, eps = self.cfg.beta_1, self.cfg.beta_2, self.cfg.epsilon
        self.momentum_buffer[:] = b1 * self.momentum_buffer + (1 - b1) * grads           # optim.py:31-34
        self.v_buffer[:] = b2 * self.v_buffer + (1 - b2) * (grads * grads)
        self.iter += 1
        m_cap = self.momentum_buffer / (1 - b1 ** self.iter)                              # bias-corrected (36-37)
        v_cap = self.v_buffer / (1 - b2 ** self.iter)
        return parameters - (self.lr * m_cap) / (np.sqrt(v_cap) + eps)


class Momentum(Optimizer):
    """default_config.py:29 names a 'Momentum' type that the reference never defines; plain heavy-ball."""

    def initialize(self):
        self.buffer = np.zeros(self.parameters_shape, dtype=np.float64)

    def _step(self, parameters, grads):
        self.buffer[:] = self.cfg.momentum * self.buffer + grads
        return parameters - self.lr * self.buffer


OPTIMIZERS = {'Adam': Adam, 'Momentum': Momentum}
