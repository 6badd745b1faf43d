"""Injector -- emits `flux` pool particles per substep around the effector (fluidlab/fluidengine/effectors/injector.py).

Injector.act (injector.py:80-105) runs inside the engine's p2g launch; this class owns the parameters, the
random vectors (drawn from the global numpy RNG exactly as injector.py:54-60) and the act_range."""
import numpy as np

from fluidlab_amd import _capi
from fluidlab_amd.configs.macros import DTYPE_NP
from fluidlab_amd.utils.misc import eval_str
from .effector import Effector


class Injector(Effector):
    abi_type = _capi.FE_EFF_INJECTOR
    has_act_id = True                  # checkpoint payload carries 'act_id' (injector.py:131-171)

    def __init__(self, radius=1.0, flux=1, inject_v=(0.0, 0.0, 0.0), inject_p=(0.0, 0.0, 0.0), randomize_inject_v=False,
                 locally_random=False, **kwargs):
        super().__init__(**kwargs)
        self.radius = radius
        self.n_particles = flux
        self.locally_random = locally_random
        self.randomize_inject_v = randomize_inject_v
        self.inject_v = tuple(eval_str(inject_v))
        self.inject_p = tuple(eval_str(inject_p))
        self.has_dynamics = False
        self.act_range = None
        self.init_random_vector()

    # tests may install a callable (random_length, flux, dim) -> array to control the injection noise
    random_vector_factory = None

    def init_random_vector(self):
        """injector.py:54-60.  Note the reference's `locally_random` indexes the noise by the *local* frame, so
        the noise sequence (and hence the trajectory) depends on max_substeps_local."""
        random_length = self.max_substeps_local if self.locally_random else self.max_substeps_global
        if Injector.random_vector_factory is not None:
            self.random_vector = np.asarray(Injector.random_vector_factory(random_length, self.n_particles, self.dim), dtype=DTYPE_NP)
            return
        self.random_vector = np.random.uniform(size=(random_length, self.n_particles, self.dim)).astype(DTYPE_NP)

    def _abi_desc(self, elib):
        d = super()._abi_desc(elib)
        d.update(flux=self.n_particles, radius=self.radius, inject_v=self.inject_v, inject_p=self.inject_p,
                 locally_random=self.locally_random, randomize_inject_v=self.randomize_inject_v,
                 random_vector=self.random_vector)
        return d

    def set_act_range(self, used):
        """injector.py:62-68: the pool = particles unused at build time."""
        self.act_range = np.where(used == False)[0].astype(np.int32)      # noqa: E712
        self.engine.eff_set_act_range(self.index, self.act_range)

    def get_state(self, f):
        return self.engine.eff_get_state(self.index, f)                   # 8 values: pos, quat, act_id (injector.py:190-196)

    @property
    def state_dim(self):
        return 8


class BallInjector(Injector):
    """injector.py:216-258: offsets uniform in a ball, no inject_p rotation; the engine's Injector.act adds
    (rv*2-1)*radius, so the ball offsets are mapped to that form."""

    def init_random_vector(self):
        random_length = self.max_substeps_local if self.locally_random else self.max_substeps_global
        need = self.n_particles * random_length
        chunks, n_generated = [], 0
        while True:
            rand_pos = np.random.uniform(high=self.radius, low=-self.radius, size=(need, 3))
            rand_pos = rand_pos[np.linalg.norm(rand_pos, axis=1) <= self.radius]
            n_generated += rand_pos.shape[0]
            chunks.append(rand_pos)
            if n_generated >= need:
                break
        offsets = np.concatenate(chunks)[:need].reshape([random_length, self.n_particles, 3])
        # offset = (rv*2-1)*radius  <=>  rv = (offset/radius + 1)/2
        self.random_vector = ((offsets / self.radius + 1.0) * 0.5).astype(DTYPE_NP)
