"""A yaml-backed stand-in for yacs.CfgNode (yacs is not in this image): attribute access over nested
dicts, defaults of fluidlab/configs/default_config.py, load_config() of fluidlab/utils/config.py:54-59."""
import ast
import copy
import os

import yaml

from fluidlab_amd.utils.misc import get_src_dir


class CfgNode(dict):
    def __init__(self, init=None, new_allowed=True):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode._coerce(v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def _coerce(v):
        """yaml leaves '(0.5, 0.5)' and '1e-3' as strings; the reference eval()s them lazily (misc.py:20-24)."""
        if isinstance(v, dict):
            return CfgNode(v)
        if isinstance(v, list):
            return [CfgNode._coerce(t) for t in v]
        if isinstance(v, str):
            try:
                return ast.literal_eval(v)
            except (ValueError, SyntaxError):
                try:
                    return float(v)
                except ValueError:
                    return v
        return v

    def merge_from_dict(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge_from_dict(v)
            else:
                self[k] = CfgNode._coerce(v)

    def merge_from_file(self, path):
        with open(path) as f:
            self.merge_from_dict(yaml.safe_load(f) or {})

    def clone(self):
        return copy.deepcopy(self)


def get_default_cfg():
    """default_config.py:3-38"""
    return CfgNode({
        'EXP': {'seed': 0, 'env_name': ''},
        'SOLVER': {'n_iters': 100, 'init_range': {'v': (), 'p': ()}, 'init_sampler': 'uniform',
                   'optim': {'lr': 0.1, 'bounds': (-1.0, 1.0), 'type': '', 'beta_1': 0.9, 'beta_2': 0.9999,
                             'epsilon': 1e-8, 'momentum': 0.9, 'trainable': None, 'fix_dim': None}},
    })


def load_config(cfg_file_name=None):
    cfg = get_default_cfg()
    if cfg_file_name is not None:
        path = cfg_file_name if os.path.isabs(cfg_file_name) else os.path.join(get_src_dir(), cfg_file_name)
        cfg.merge_from_file(path)
    return cfg
