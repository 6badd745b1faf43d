"""IceCreamStaticLoss (fluidlab/fluidengine/losses/icecreamstatic_loss.py): shape matching of the ICECREAM1 particles."""
import numpy as np

from fluidlab_amd.configs.macros import ICECREAM1
from .shapematching_loss import ShapeMatchingLoss


class IceCreamStaticLoss(ShapeMatchingLoss):
    def __init__(self, type, **kwargs):
        if type == 'diff':
            super().__init__(matching_mat=ICECREAM1, temporal_init_range_end=100, temporal_range_type='expand', **kwargs)
        elif type == 'default':
            super().__init__(matching_mat=ICECREAM1, temporal_range_type='all', **kwargs)
        else:
            raise ValueError(type)

    def get_step_loss(self):
        cur_step_loss = float(self.step_loss[self.sim.cur_step_global - 1])
        return {'reward': 0.001 * (900 - cur_step_loss), 'loss': cur_step_loss}            # icecreamstatic_loss.py:32-40

    def get_final_loss(self):
        info = super().get_final_loss()
        info['reward'] = float(np.sum((900 - self.step_loss) * 0.001))
        return info
