"""HostLoss -- shared plumbing of the task losses whose per-step term is a small reduction over particle positions
(gatheringeasy / gatheringO / pouring / transporting / mixing _loss.py in fluidlab/fluidengine/losses).  The engine owns the
frames; a step reads x/used of one frame, the subclass returns the step's value and d value / d x, and the backward step hands
the adjoint back.  The arithmetic is written once against a small array namespace (`xp`): on the HIP engine it runs as torch
ops on the engine's GPU, fed through the device-pointer entry points (`fe_get_frame_dev` / `fe_add_grad_dev`: the frame never
crosses PCIe); against the CPU oracle (tests) it is numpy on `fe_get_frame` / `fe_add_grad`.  Temporal-range handling ('last' /
'all' / 'expand' with the plateau rule) is the block every one of those reference files repeats."""
import numpy as np

from .loss import Loss


SMALL_ON_HOST = 20000        # pairwise_l1: below this many points the sort-based sum runs on the host even for GPU-resident frames


class _NumpyOps:
    """float64 numpy"""
    name = 'numpy'

    def f64(self, a): return np.asarray(a, np.float64)
    def asarray(self, a, like=None): return np.asarray(a)
    def zeros_like(self, a): return np.zeros_like(a)
    def zeros(self, shape, like): return np.zeros(shape, np.float64)
    def abs(self, a): return np.abs(a)
    def sign(self, a): return np.sign(a)
    def sort(self, a): return np.sort(a)
    def cumsum0(self, a): return np.concatenate([[0.0], np.cumsum(a)])
    def searchsorted(self, s, v, side): return np.searchsorted(s, v, side)
    def where(self, m): return np.where(m)[0]
    def norm_rows(self, a): return np.linalg.norm(a, axis=1)
    def argmin(self, a): return int(np.argmin(a))
    def rank(self, a): return np.argsort(np.argsort(a))
    def to_float(self, a): return float(a)
    def count(self, m): return int(np.count_nonzero(m))


class _TorchOps:
    """float64 torch tensors on the engine's GPU"""
    name = 'torch'

    def __init__(self, device):
        import torch
        self.t = torch
        self.device = device

    def f64(self, a): return a.to(self.t.float64)
    def asarray(self, a, like=None): return a if self.t.is_tensor(a) else self.t.as_tensor(np.asarray(a), device=self.device)
    def zeros_like(self, a): return self.t.zeros_like(a)
    def zeros(self, shape, like): return self.t.zeros(shape, dtype=self.t.float64, device=self.device)
    def abs(self, a): return a.abs()
    def sign(self, a): return a.sign()
    def sort(self, a): return self.t.sort(a).values
    def cumsum0(self, a): return self.t.cat([self.t.zeros(1, dtype=a.dtype, device=a.device), self.t.cumsum(a, 0)])
    def searchsorted(self, s, v, side): return self.t.searchsorted(s, v.contiguous(), right=(side == 'right'))
    def where(self, m): return self.t.nonzero(m, as_tuple=False)[:, 0]
    def norm_rows(self, a): return self.t.linalg.norm(a, dim=1)
    def argmin(self, a): return int(self.t.argmin(a))
    def rank(self, a): return self.t.argsort(self.t.argsort(a))
    def to_float(self, a): return float(a)
    def count(self, m): return int(m.sum())


def pairwise_l1(a, b=None, xp=None):
    """sum_{i,j} |a_i - b_j|_1 and its gradients without forming the pairs: per dimension the sum separates after a sort
    (value by prefix sums, d/d a_i = #{b < a_i} - #{b > a_i}).  b=None: all ordered pairs (i, j) of `a` itself
    (mixing_loss.py:74-75), where every unordered pair appears twice.  float64 in and out."""
    xp = xp or _NumpyOps()
    if xp.name == 'torch' and len(a) + (0 if b is None else len(b)) < SMALL_ON_HOST:
        # a few thousand points: a dozen tiny GPU launches per dimension cost more than one small D2H copy and numpy
        t = xp.t
        total, ga, gb = pairwise_l1(a.detach().cpu().numpy(), None if b is None else b.detach().cpu().numpy())
        return total, t.as_tensor(ga, device=a.device), (None if gb is None else t.as_tensor(gb, device=a.device))
    a = xp.f64(xp.asarray(a))
    self_pairs = b is None
    b = a if self_pairs else xp.f64(xp.asarray(b))
    total, ga, gb = 0.0, xp.zeros_like(a), xp.zeros_like(b)
    if len(a) == 0 or len(b) == 0:
        return total, ga, (None if self_pairs else gb)
    for d in range(a.shape[1]):
        ad, bd = a[:, d], b[:, d]
        bs = xp.sort(bd)
        csum = xp.cumsum0(bs)
        lo, hi = xp.searchsorted(bs, ad, 'left'), xp.searchsorted(bs, ad, 'right')
        # b < a_i : lo of them (their sum csum[lo]); b > a_i : len - hi of them
        total += xp.to_float((ad * lo - csum[lo]).sum() + ((csum[-1] - csum[hi]) - ad * (len(bs) - hi)).sum())
        ga[:, d] = lo - (len(bs) - hi)
        if not self_pairs:
            as_ = xp.sort(ad)
            lo_b, hi_b = xp.searchsorted(as_, bd, 'left'), xp.searchsorted(as_, bd, 'right')
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
        self.xp = _NumpyOps()
        if self.engine.elib.backend.startswith('hip'):           # frames stay in HBM: evaluate there
            import torch
            self._dev = torch.device('cuda', self.engine.device)
            self.xp = _TorchOps(self._dev)
            self._x_dev = torch.zeros((self.n_particles, 3), dtype=torch.float32, device=self._dev)
            self._used_dev = torch.zeros((self.n_particles,), dtype=torch.int32, device=self._dev)

    @property
    def step_loss(self):
        return self._step_loss

    @property
    def particle_mat(self):
        """material ids, as an array of the active namespace"""
        if self._mat is None:
            self._mat = self.xp.asarray(self.sim.particles_i.mat.to_numpy())
        return self._mat

    def clear_loss(self):
        super().clear_loss()
        if hasattr(self, '_step_loss'):
            self._step_loss[:] = 0
            self.total_loss = 0.0

    def frame(self, f):
        """x [N,3] (float64) and used [N] (bool) of frame f in the active namespace"""
        if self.xp.name == 'torch':
            self.engine.get_frame_dev(f, x=self._x_dev, used=self._used_dev)         # returns after the engine's stream has finished
            return self._x_dev.to(self.xp.t.float64), self._used_dev > 0
        x = np.zeros((self.n_particles, 3), self.engine.dtype); used = np.zeros((self.n_particles,), np.int32)
        self.engine.get_frame(f, x=x, used=used)
        return x.astype(np.float64), used > 0

    # subclasses: value of step s at frame f (float) and, when asked, d value / d x as an [N, 3] array (or None)
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
        if gx is None:
            return
        if self.xp.name == 'torch':
            g32 = (gx * self.total_loss_grad).to(self.xp.t.float32).contiguous()
            self.xp.t.cuda.synchronize(self._dev)                 # the engine reads it on its own stream
            self.engine.add_grad_dev(f, gx=g32)
        else:
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
