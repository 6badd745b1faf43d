"""Domain / effector boundaries (fluidlab/fluidengine/boundaries/boundaries.py).

In the reference these are @ti.func bodies inlined into grid_op and Effector.move_kernel by the
Taichi JIT.  Here a boundary is a parameter record: the arithmetic lives in csrc/fe_math.h
(boundary_v / boundary_x) and the record is handed to the engine as an FeBoundary struct."""
import numpy as np

from fluidlab_amd.configs.macros import DTYPE_NP
from fluidlab_amd.utils.misc import eval_str


class Boundary:
    type = None

    def __init__(self, restitution=0.0, lock_dims=()):
        self.restitution = float(restitution)
        self.lock_dims = [int(d) for d in lock_dims]

    def abi_kwargs(self):
        raise NotImplementedError

    def to_abi(self, elib):
        return elib.make_boundary(**self.abi_kwargs())

    # host-side evaluation, used by policies / tests (same rules as the kernels)
    def impose_x(self, x):
        raise NotImplementedError

    def is_out(self, x):
        raise NotImplementedError


class CylinderBoundary(Boundary):
    """boundaries.py:28-93"""
    type = 'cylinder'

    def __init__(self, y_range=(0.05, 0.95), xz_center=(0.5, 0.5), xz_radius=0.45, **kwargs):
        super().__init__(**kwargs)
        self.y_range = np.array(eval_str(y_range), dtype=DTYPE_NP)
        self.xz_center = np.array(eval_str(xz_center), dtype=DTYPE_NP)
        self.xz_radius = float(xz_radius)

    def abi_kwargs(self):
        return dict(type='cylinder', y_range=self.y_range, xz_center=self.xz_center, xz_radius=self.xz_radius,
                    restitution=self.restitution, lock_dims=self.lock_dims)

    def impose_x(self, x):
        x = np.array(x, dtype=DTYPE_NP)
        out = x.copy()
        out[1] = min(max(x[1], self.y_range[0]), self.y_range[1])
        r = x[[0, 2]] - self.xz_center
        nrm = np.sqrt((r * r).sum() + 1e-12)
        if nrm > self.xz_radius:
            out[[0, 2]] = r / nrm * self.xz_radius + self.xz_center
        return out

    def is_out(self, x):
        r = np.asarray(x)[[0, 2]] - self.xz_center
        return bool(x[1] > self.y_range[1] or x[1] < self.y_range[0] or np.sqrt((r * r).sum() + 1e-12) > self.xz_radius)


class CubeBoundary(Boundary):
    """boundaries.py:96-134"""
    type = 'cube'

    def __init__(self, lower=(0.05, 0.05, 0.05), upper=(0.95, 0.95, 0.95), **kwargs):
        super().__init__(**kwargs)
        self.upper = np.array(eval_str(upper), dtype=DTYPE_NP)
        self.lower = np.array(eval_str(lower), dtype=DTYPE_NP)
        assert (self.upper >= self.lower).all()

    def abi_kwargs(self):
        return dict(type='cube', lower=self.lower, upper=self.upper, restitution=self.restitution, lock_dims=self.lock_dims)

    def impose_x(self, x):
        return np.maximum(np.minimum(np.asarray(x, dtype=DTYPE_NP), self.upper), self.lower)

    def is_out(self, x):
        x = np.asarray(x)
        return bool((x > self.upper).any() or (x < self.lower).any())


def create_boundary(type='cube', **kwargs):
    """boundaries.py:136-141"""
    if type == 'cylinder':
        return CylinderBoundary(**kwargs)
    if type == 'cube':
        return CubeBoundary(**kwargs)
    raise AssertionError(f'unknown boundary type: {type}')
