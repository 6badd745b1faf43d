"""IceCreamDynamicLoss (fluidlab/fluidengine/losses/icecreamdynamic_loss.py): shape matching of the ICECREAM particles,
expanding temporal range for the differentiable solver."""
import numpy as np

from fluidlab_amd.configs.macros import ICECREAM
from .shapematching_loss import ShapeMatchingLoss


class IceCreamDynamicLoss(ShapeMatchingLoss):
    def __init__(self, type, **kwargs):
        if type == 'diff':
            super().__init__(matching_mat=ICECREAM, temporal_init_range_end=200, temporal_range_type='expand', **kwargs)
        elif type == 'default':
            super().__init__(matching_mat=ICECREAM, temporal_range_type='all', **kwargs)
        else:
            raise ValueError(type)

    def get_step_loss(self):
        cur_step_loss = float(self.step_loss[self.sim.cur_step_global - 1])
        return {'reward': 0.001 * (1700 - cur_step_loss), 'loss': 0.001 * cur_step_loss}      # icecreamdynamic_loss.py:33-41

    def get_final_loss(self):
        info = super().get_final_loss()
        info['reward'] = float(np.sum((1700 - self.step_loss) * 0.001))
        return info
