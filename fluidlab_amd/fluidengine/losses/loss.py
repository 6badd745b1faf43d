"""Loss base class (fluidlab/fluidengine/losses/loss.py).  step_loss / its adjoint live on the host as numpy
arrays; the per-particle reductions and the adjoint write into particles.grad.x run in the engine."""
import numpy as np


class Loss:
    def __init__(self, max_loss_steps, weights=None, target_file=None):
        self.weights = weights
        self.target_file = target_file
        self.inf = 1e8
        self.max_loss_steps = max_loss_steps
        self.step_loss_grad = np.zeros((max_loss_steps,), np.float32)      # step_loss.grad (loss.py:21)
        self.total_loss_grad = 1.0                                         # total_loss.grad
        self._step_loss_cache = None

    def build(self, sim):
        self.sim = sim
        self.engine = sim.engine
        self.res, self.n_grid, self.dx, self.dim = sim.res, sim.n_grid, sim.dx, sim.dim
        if sim.agent is not None:
            self.agent = sim.agent
        if sim.particles is not None:
            self.n_particles = sim.n_particles
        self.engine.loss_alloc(self.max_loss_steps)
        if self.target_file is not None:
            self.load_target(self.target_file)
        self.reset()

    def reset_grad(self):
        self.step_loss_grad[:] = 0          # loss.py:46-48
        self.total_loss_grad = 1.0

    def load_target(self, path):
        pass

    def clear_loss(self):
        self.engine.loss_clear()            # loss.py:54-61
        self.step_loss_grad[:] = 0
        self.total_loss_grad = 1.0
        self._step_loss_cache = None

    def clear_losses(self):
        pass

    def reset(self):
        self.clear_loss()
        self.clear_losses()

    @property
    def step_loss(self):
        """step_loss[...] as a host array (one D2H read, cached until the next loss kernel)."""
        if self._step_loss_cache is None:
            self._step_loss_cache = self.engine.loss_get(self.max_loss_steps)
        return self._step_loss_cache

    def step(self):
        self._step_loss_cache = None
        self.compute_step_loss(self.sim.cur_step_global - 1, self.sim.cur_substep_local)          # loss.py:72-74

    def step_grad(self):
        self.compute_step_loss_grad(self.sim.cur_step_global - 1, self.sim.cur_substep_local)     # loss.py:76-78
