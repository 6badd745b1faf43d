"""Open-loop action policies for the trajectory optimiser.

Contract kept from the reference (fluidlab/optimizer/policies.py; consumed by Solver and the envs): a policy exposes
`get_actions_p()`, `get_action_v(i, agent=None, update=False)`, `comp_actions` = [actions_v (horizon x dim); actions_p],
`optimize(grads, loss_info)`, and the attributes `trainable`, `status`, `freeze_till`, `actions_v`, `actions_p`, `optim`.
The task-specific behaviour of the reference's subclasses is expressed here as data -- a periodic *stage plan*
(which steps of a cycle are optimised, which are scripted and how), a learning-rate ladder and a freeze rule keyed on
the loss' temporal range -- interpreted by one class.  Interactive (keyboard / mouse) policies need a display and are
not part of this package.  tests/test_host_golden.py pins masks and one optimisation step against the reference.
"""
import numpy as np

from .optim import OPTIMIZERS

# stage kinds of a plan: what get_action_v(update=True) writes for a step of that stage
TRAIN, LIFT, RETURN, LOWER = 0, 1, 2, 3


class ActionsPolicy:
    """Replays a recorded action table: rows 0..H-1 are per-step velocities, the last row the initial pose (policies.py:10-19)."""

    def __init__(self, comp_actions):
        table = np.asarray(comp_actions)
        self.actions_v, self.actions_p = table[:-1], table[-1]

    def get_actions_p(self):
        return self.actions_p

    def get_action_v(self, i, **kwargs):
        return self.actions_v[i]


class StagePlan:
    """A cycle of `period` steps cut into stages [(end_step, kind), ...]; step i is in the first stage whose end exceeds
    i % period.  TRAIN steps are optimised, the others scripted: LIFT / LOWER move by +/- `dy` in y, RETURN heads for
    `target` (None: the policy's initial pose, with y held) so as to arrive when the stage ends."""

    def __init__(self, stages, dy=0.008, target=None):
        self.stages = list(stages)
        self.period = self.stages[-1][0]
        self.dy = dy
        self.target = None if target is None else np.asarray(target, dtype=np.float64)

    def stage_of(self, i):
        r = i % self.period
        for k, (end, kind) in enumerate(self.stages):
            if r < end:
                return k, kind
        raise AssertionError('stage plan does not cover its period')

    def masks(self, horizon, n_rows):
        trainable = np.zeros(n_rows, dtype=bool)
        status = np.zeros(n_rows, dtype=np.int64)
        for i in range(horizon):
            k, kind = self.stage_of(i)
            status[i] = k
            trainable[i] = kind == TRAIN
        return trainable, status

    def scripted(self, i, start_pose, latest_pos):
        """action of a scripted step, or None for a TRAIN step"""
        k, kind = self.stage_of(i)
        if kind == TRAIN:
            return None
        if kind in (LIFT, LOWER):
            return np.array([0.0, self.dy if kind == LIFT else -self.dy, 0.0])
        remaining = self.stages[k][0] - i % self.period
        if self.target is None:
            step = (start_pose - latest_pos) / remaining
            step[1] = 0
            return step
        return (self.target - latest_pos) / remaining


class TrainablePolicy:
    """The optimised action sequence (policies.py:131-164): `horizon` velocity rows + one initial-pose row, masked Adam."""

    plan = None                 # StagePlan of the task, if its horizon is staged
    head_frozen = 0.0           # fraction of the horizon at the start that is never optimised (the demo's hold-still phase)
    pose_trainable = True       # is the initial-pose row optimised
    only_velocity_rows = False  # start from "every velocity row, not the pose row"
    lr_ladder = ()              # ((temporal_range_above, lr_factor), ...) checked in order after each update
    grad_clip = None
    # freeze rule: (thresholds descending, lag, when); when = 'before' | 'after' the update
    freeze = None

    def __init__(self, optim_cfg, init_range, action_dim, horizon, action_range, fix_dim=None):
        self.horizon, self.action_dim = horizon, action_dim
        self.actions_v = np.random.uniform(init_range.v[0], init_range.v[1], size=(horizon, action_dim))
        self.actions_p = np.random.uniform(init_range.p[0], init_range.p[1], size=(action_dim))
        self.action_range = action_range
        self.comp_actions_shape = (horizon + 1, action_dim)
        self.fix_dim = fix_dim
        self.freeze_till = 0
        self.optim = OPTIMIZERS[optim_cfg.type](self.comp_actions_shape, optim_cfg)
        self.trainable, status = self._initial_masks()
        if status is not None:
            self.status = status
            self.stage_step = [end for end, _ in self.plan.stages]

    def _initial_masks(self):
        rows = self.comp_actions_shape[0]
        if self.plan is not None:
            return self.plan.masks(self.horizon, rows)
        mask = np.ones(rows, dtype=bool)
        if self.only_velocity_rows or self.head_frozen or not self.pose_trainable:
            mask[:] = False
            mask[int(round(self.head_frozen * self.horizon)):-1] = True
        return mask, None

    @property
    def comp_actions(self):
        return np.vstack([self.actions_v, self.actions_p[None, :]])

    def get_actions_p(self):
        return self.actions_p

    def get_action_v(self, i, agent=None, update=False):
        if update and self.plan is not None:
            act = self.plan.scripted(i, self.actions_p, None if self.plan.stage_of(i)[1] != RETURN else agent.rigid.latest_pos.to_numpy()[0])
            if act is not None:
                self.actions_v[i] = act
        return self.actions_v[i]

    def _apply_freeze(self, temporal_range):
        thresholds, lag = self.freeze[0], self.freeze[1]
        for t in thresholds:
            if temporal_range > t:
                self.freeze_till = temporal_range - lag
                self.trainable[:max(self.freeze_till, 0)] = False
                break

    def optimize(self, grads, loss_info):
        assert grads.shape == self.comp_actions_shape
        if self.freeze is not None and self.freeze[2] == 'before':
            self._apply_freeze(loss_info['temporal_range'])
        g = np.array(grads, dtype=np.float64)
        if self.grad_clip is not None:
            g = np.clip(g, -self.grad_clip, self.grad_clip)
        g[~self.trainable] = 0
        if self.fix_dim is not None:
            g[:, self.fix_dim] = 0
        table = self.optim.step(self.comp_actions, g)
        self.actions_p = table[-1]
        self.actions_v = table[:-1].clip(*self.action_range)
        for above, factor in self.lr_ladder:
            if loss_info['temporal_range'] > above:
                self.optim.lr = self.optim.init_lr * factor
                break
        if self.freeze is not None and self.freeze[2] == 'after':
            self._apply_freeze(loss_info['temporal_range'])


class LatteArtPolicy(TrainablePolicy):
    """policies.py:167-169: nothing task-specific"""


class CirculationPolicy(TrainablePolicy):
    """policies.py:341-347"""


class PouringPolicy(TrainablePolicy):
    """policies.py:357-359"""


class LatteArtStirPolicy(TrainablePolicy):
    """policies.py:172-194: the step size drops as the loss' temporal range grows, and the early part of the stir is frozen
    progressively (only `trainable` is touched, not `freeze_till`, as in the reference)."""
    lr_ladder = ((250, 0.2), (150, 0.5))

    def optimize(self, grads, loss_info):
        super().optimize(grads, loss_info)
        for step in (400, 350, 300, 250, 200, 150, 100):
            if loss_info['temporal_range'] > step:
                self.trainable[:step - 100] = False
                break


class IceCreamDynamicPolicy(TrainablePolicy):
    """policies.py:196-201: the demo's hold-still phase (168 of 900 steps) and the initial pose are not optimised"""
    head_frozen = 169 / 900
    pose_trainable = False


class IceCreamStaticPolicy(TrainablePolicy):
    """policies.py:204-216"""
    only_velocity_rows = True
    grad_clip = 1e5
    lr_ladder = ((450, 0.1),)


class TransportingPolicy(TrainablePolicy):
    """policies.py:362-366: the initial pose is not optimised"""
    only_velocity_rows = True


class GatheringPolicy(TrainablePolicy):
    """policies.py:218-259: sweeps of 120 steps -- 50 optimised pushing steps, 15 up, 40 back over the start, 15 down; sweeps
    that the loss' temporal range has left more than one sweep behind are frozen before the update."""
    plan = StagePlan([(50, TRAIN), (65, LIFT), (105, RETURN), (120, LOWER)])
    freeze = ((720, 600, 480, 360, 240, 120), 120, 'before')


class GatheringOPolicy(GatheringPolicy):
    """policies.py:262-303: the same sweeps without the freezing"""
    freeze = None


class MixingPolicy(TrainablePolicy):
    """policies.py:306-338: cycles of 80 steps -- 50 optimised stirring steps, 30 back to the rest pose above the cup; cycles
    more than two behind the loss' temporal range are frozen after the update."""
    rest_pos = np.array([0.5, 0.73, 0.5])
    plan = StagePlan([(50, TRAIN), (80, RETURN)], target=rest_pos)
    freeze = (tuple(range(80, 2000, 80))[::-1], 160, 'after')
