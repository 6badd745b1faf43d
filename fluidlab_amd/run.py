"""CLI (fluidlab/run.py): record a target with the demo policy, or optimise a trajectory.

  python -m fluidlab_amd.run --cfg_file configs/exp_latteart.yaml --record
  python -m fluidlab_amd.run --cfg_file configs/exp_latteart.yaml --exp_name latte
Multi-GPU (one env replica per GPU, action gradients all-reduced over RCCL):
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m fluidlab_amd.run --cfg_file ...
"""
import argparse
import os
import pickle as pkl

import fluidlab_amd.envs as envs
from fluidlab_amd.optimizer.distributed import EnvParallel
from fluidlab_amd.optimizer.recorder import record_target, replay_policy
from fluidlab_amd.optimizer.solver import solve_policy
from fluidlab_amd.utils.config import load_config
from fluidlab_amd.utils.misc import get_src_dir


class Logger:
    """Policy pickles + stdout (fluidlab/utils/logger.py without TensorBoard / image writers)."""

    def __init__(self, exp_name):
        self.dir = os.path.join(get_src_dir(), '..', 'logs', 'policies', exp_name)
        os.makedirs(self.dir, exist_ok=True)

    def save_policy(self, policy, iteration):
        with open(os.path.join(self.dir, f'{iteration:04d}.pkl'), 'wb') as fh:
            pkl.dump(policy, fh)

    def log(self, iteration, info):
        print(f'Iteration: {iteration}, ' + ', '.join(f'{k}: {v:.4f}' if isinstance(v, float) else f'{k}: {v}' for k, v in info.items()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--exp_name', type=str, default='test')
    ap.add_argument('--env_name', type=str, default='')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--cfg_file', type=str, default=None)
    ap.add_argument('--record', action='store_true')
    ap.add_argument('--replay_policy', action='store_true')
    ap.add_argument('--path', type=str, default=None)
    args = ap.parse_args()
    cfg = load_config(args.cfg_file) if args.cfg_file is not None else None
    env_name = cfg.EXP.env_name if cfg is not None else args.env_name
    seed = cfg.EXP.seed if cfg is not None else args.seed
    par = EnvParallel()
    if args.record:
        record_target(envs.make(env_name, seed=seed, loss=False, loss_type='diff', device=par.local_rank))
    elif args.replay_policy:
        replay_policy(envs.make(env_name, seed=seed, loss=False, loss_type='diff', device=par.local_rank), path=args.path)
    else:
        # replicas differ by their injector randomness: seed = base seed + rank
        env = envs.make(env_name, seed=seed + par.rank, loss=True, loss_type='diff', device=par.local_rank)
        solve_policy(env, Logger(args.exp_name) if par.rank == 0 else None, cfg.SOLVER, parallel=par if par.world_size > 1 else None)
    par.close()


if __name__ == '__main__':
    main()
