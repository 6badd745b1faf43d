"""Small host helpers (fluidlab/utils/misc.py)."""
import ast
import os
import random
import socket

import numpy as np

import fluidlab_amd


def get_src_dir():
    return os.path.dirname(fluidlab_amd.__file__)


def get_cfg_path(file):
    return os.path.join(get_src_dir(), 'envs', 'configs', file)


def get_tgt_path(file):
    return os.path.join(get_src_dir(), 'assets', 'targets', file)


def eval_str(x):
    """Config values such as '(0.5, 0.5)' arrive as strings (misc.py:20-24); literals only."""
    return ast.literal_eval(x) if isinstance(x, str) else x


def is_on_server():
    return True                     # headless: there is no renderer in this package


def set_random_seed(seed):
    """misc.py:35-39 -- the injector's random vectors come from the global numpy RNG."""
    random.seed(seed)
    np.random.seed(seed)
    try:
        import torch
        torch.manual_seed(seed)
    except ImportError:
        pass
