"""Solver -- trajectory optimisation by back-propagating through the simulation (fluidlab/optimizer/solver.py)."""
from time import time

import numpy as np


class Solver:
    def __init__(self, env, logger=None, cfg=None, parallel=None):
        self.cfg = cfg
        self.env = env
        self.target_file = env.target_file
        self.logger = logger
        self.parallel = parallel          # optimizer.distributed.EnvParallel or None

    def forward_backward(self, sim_state, policy, horizon, horizon_action):
        """One forward rollout with loss, then the reverse sweep (solver.py:23-59).
        Returns (loss_info, dLoss/d comp_actions of shape (horizon_action+1, action_dim))."""
        taichi_env = self.env.taichi_env
        taichi_env.set_state(sim_state, grad_enabled=True)
        t1 = time()
        taichi_env.apply_agent_action_p(policy.get_actions_p())
        cur_horizon = taichi_env.loss.temporal_range[1]
        for i in range(cur_horizon):
            action = policy.get_action_v(i, agent=taichi_env.agent, update=True) if i < horizon_action else None
            taichi_env.step(action)
        loss_info = taichi_env.get_final_loss()
        t2 = time()
        taichi_env.reset_grad()
        taichi_env.get_final_loss_grad()
        for i in range(cur_horizon - 1, policy.freeze_till - 1, -1):
            action = policy.get_action_v(i) if i < horizon_action else None
            taichi_env.step_grad(action)
        taichi_env.apply_agent_action_p_grad(policy.get_actions_p())
        grad = taichi_env.agent.get_grad(horizon_action)
        t3 = time()
        loss_info['forward_s'], loss_info['backward_s'] = t2 - t1, t3 - t2
        print(f'=======> forward: {t2 - t1:.2f}s backward: {t3 - t2:.2f}s')
        return loss_info, grad

    def solve(self, policy=None, callback=None):
        taichi_env = self.env.taichi_env
        if policy is None:
            policy = self.env.trainable_policy(self.cfg.optim, self.cfg.init_range)
        init_state = taichi_env.get_state()
        for iteration in range(self.cfg.n_iters):
            if self.logger is not None:
                self.logger.save_policy(policy, iteration)
            loss_info, grad = self.forward_backward(init_state['state'], policy, self.env.horizon, self.env.horizon_action)
            if self.parallel is not None:
                grad, (loss_mean,) = self.parallel.all_reduce_mean(grad, [loss_info['loss']])
                loss_info['loss_mean_over_envs'] = float(loss_mean)
            loss_info['iteration'] = iteration
            policy.optimize(grad, loss_info)
            if self.logger is not None:
                loss_info['lr'] = policy.optim.lr
                self.logger.log(iteration, loss_info)
            if callback is not None:
                callback(iteration, loss_info, policy)
        return policy


def solve_policy(env, logger, cfg, parallel=None):
    env.reset()
    return Solver(env, logger, cfg, parallel=parallel).solve()
