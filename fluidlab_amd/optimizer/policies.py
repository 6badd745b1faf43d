"""Open-loop action policies (fluidlab/optimizer/policies.py): replayed action lists and the trainable
action sequence optimised by Solver.  Interactive keyboard/mouse policies need a display and are omitted."""
import numpy as np

from .optim import OPTIMIZERS


class ActionsPolicy:
    """comp_actions = [actions_v (horizon x dim) ; actions_p (1 x dim)]  (policies.py:10-19)"""

    def __init__(self, comp_actions):
        self.actions_v = comp_actions[:-1]
        self.actions_p = comp_actions[-1]

    def get_actions_p(self):
        return self.actions_p

    def get_action_v(self, i, **kwargs):
        return self.actions_v[i]


class TrainablePolicy:
    """policies.py:131-164"""

    def __init__(self, optim_cfg, init_range, action_dim, horizon, action_range, fix_dim=None):
        self.horizon = horizon
        self.action_dim = action_dim
        self.actions_v = np.random.uniform(init_range.v[0], init_range.v[1], size=(horizon, action_dim))
        self.actions_p = np.random.uniform(init_range.p[0], init_range.p[1], size=(action_dim))
        self.action_range = action_range
        self.comp_actions_shape = (horizon + 1, action_dim)
        self.trainable = np.full(self.comp_actions_shape[0], True)
        self.fix_dim = fix_dim
        self.freeze_till = 0
        self.optim = OPTIMIZERS[optim_cfg.type](self.comp_actions_shape, optim_cfg)

    @property
    def comp_actions(self):
        return np.vstack([self.actions_v, self.actions_p[None, :]])

    def get_actions_p(self):
        return self.actions_p

    def get_action_v(self, i, **kwargs):
        return self.actions_v[i]

    def optimize(self, grads, loss_info):
        assert grads.shape == self.comp_actions_shape
        grads = np.array(grads, dtype=np.float64)
        grads[np.logical_not(self.trainable)] = 0
        if self.fix_dim is not None:
            grads[:, self.fix_dim] = 0
        new_comp_actions = self.optim.step(self.comp_actions, grads)
        self.actions_p = new_comp_actions[-1]
        self.actions_v = new_comp_actions[:-1].clip(*self.action_range)


class LatteArtPolicy(TrainablePolicy):
    pass


class GatheringPolicy(TrainablePolicy):
    """policies.py:218-259: a repeating sweep of 120 steps -- 50 trainable steps pushing, 15 up, 40 back to the start, 15 down."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.trainable = np.full(self.comp_actions_shape[0], False)
        self.status = np.full(self.comp_actions_shape[0], 0)
        self.stage_step = [50, 65, 105, 120]
        for i in range(self.horizon):
            r = i % self.stage_step[3]
            if r < self.stage_step[0]:
                self.trainable[i] = True
                self.status[i] = 0          # moving
            elif r < self.stage_step[1]:
                self.status[i] = 1          # up
            elif r < self.stage_step[2]:
                self.status[i] = 2          # moving back
            else:
                self.status[i] = 3          # down

    def get_action_v(self, i, agent=None, update=False):
        if update:
            if self.status[i] == 1:
                self.actions_v[i] = np.array([0, 0.008, 0])
            elif self.status[i] == 2:
                action = (self.actions_p - agent.rigid.latest_pos.to_numpy()[0]) / (self.stage_step[2] - (i % self.stage_step[3]))
                action[1] = 0
                self.actions_v[i] = action
            elif self.status[i] == 3:
                self.actions_v[i] = np.array([0, -0.008, 0])
        return self.actions_v[i]

    def optimize(self, grads, loss_info):
        for step in [720, 600, 480, 360, 240, 120]:
            if loss_info['temporal_range'] > step:
                self.freeze_till = loss_info['temporal_range'] - 120
                self.trainable[:self.freeze_till] = False
                break
        super().optimize(grads, loss_info)


class LatteArtStirPolicy(TrainablePolicy):
    """policies.py:172-187"""

    def optimize(self, grads, loss_info):
        super().optimize(grads, loss_info)
        if loss_info['temporal_range'] > 250:
            self.optim.lr = self.optim.init_lr * 0.2
        elif loss_info['temporal_range'] > 150:
            self.optim.lr = self.optim.init_lr * 0.5


class CirculationPolicy(TrainablePolicy):
    """policies.py:341-347"""


class IceCreamDynamicPolicy(TrainablePolicy):
    """policies.py:196-201"""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.trainable = np.full(self.comp_actions_shape[0], False)
        first = int(round(169 * self.horizon / 900))      # the demo's hold-still phase (168 of 900 steps) is not optimised
        self.trainable[first:-1] = True


class IceCreamStaticPolicy(TrainablePolicy):
    """policies.py:204-216"""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.trainable = np.full(self.comp_actions_shape[0], False)
        self.trainable[:-1] = True

    def optimize(self, grads, loss_info):
        super().optimize(np.clip(grads, -1e5, 1e5), loss_info)
        if loss_info['temporal_range'] > 450:
            self.optim.lr = self.optim.init_lr * 0.1


class GatheringOPolicy(GatheringPolicy):
    """policies.py:262-303: GatheringPolicy's 120-step sweep (50 trainable steps, up, back, down) without the freezing."""

    def optimize(self, grads, loss_info):
        TrainablePolicy.optimize(self, grads, loss_info)


class MixingPolicy(TrainablePolicy):
    """policies.py:306-338: cycles of 80 steps -- 50 trainable stirring steps, then 30 steps back to the rest pose above the
    cup; once the loss' temporal range has passed a cycle boundary the cycles more than two back are frozen."""
    rest_pos = np.array([0.5, 0.73, 0.5])

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.trainable = np.full(self.comp_actions_shape[0], False)
        self.status = np.full(self.comp_actions_shape[0], 0)
        self.stage_step = [50, 80]
        for i in range(self.horizon):
            if i % self.stage_step[1] < self.stage_step[0]:
                self.trainable[i] = True
                self.status[i] = 0          # moving
            else:
                self.status[i] = 1          # moving back

    def get_action_v(self, i, agent=None, update=False):
        if update and self.status[i] == 1:
            self.actions_v[i] = (self.rest_pos - agent.rigid.latest_pos.to_numpy()[0]) / (self.stage_step[1] - (i % self.stage_step[1]))
        return self.actions_v[i]

    def optimize(self, grads, loss_info):
        super().optimize(grads, loss_info)
        for step in list(range(80, 2000, 80))[::-1]:
            if loss_info['temporal_range'] > step:
                self.freeze_till = loss_info['temporal_range'] - 160
                self.trainable[:max(self.freeze_till, 0)] = False
                break


class PouringPolicy(TrainablePolicy):
    """policies.py:357-359"""


class TransportingPolicy(TrainablePolicy):
    """policies.py:362-366: the initial pose is not optimised"""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.trainable = np.full(self.comp_actions_shape[0], False)
        self.trainable[:-1] = True
