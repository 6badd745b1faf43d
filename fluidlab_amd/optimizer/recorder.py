"""Recorder -- rolls out the demo policy and stores the particle trajectory as the loss target
(fluidlab/optimizer/recorder.py); also replays a stored policy."""
import os
import pickle as pkl


class Recorder:
    def __init__(self, env):
        self.env = env
        self.target_file = env.target_file
        if self.target_file is not None:
            os.makedirs(os.path.dirname(self.target_file), exist_ok=True)

    def record(self, user_input=False, write=True):
        policy = self.env.demo_policy(user_input)
        taichi_env = self.env.taichi_env
        init = taichi_env.get_state()
        target = {'x': [], 'used': [], 'mat': None}
        taichi_env.set_state(**init)
        action_p = policy.get_actions_p()
        if action_p is not None:
            taichi_env.apply_agent_action_p(action_p)
        for i in range(self.env.horizon):
            action = policy.get_action_v(i) if i < self.env.horizon_action else None
            taichi_env.step(action)
            if taichi_env.has_particles:
                cur = taichi_env.get_state()['state']
                target['x'].append(cur['x'])
                target['used'].append(cur['used'])
        target['mat'] = taichi_env.simulator.particles_i.mat.to_numpy()
        if write and self.target_file is not None:
            if os.path.exists(self.target_file):
                os.remove(self.target_file)
            with open(self.target_file, 'wb') as fh:
                pkl.dump(target, fh)
            print(f'===> New target generated and dumped to {self.target_file}.')
        return target

    def replay_policy(self, policy_path):
        taichi_env = self.env.taichi_env
        with open(policy_path, 'rb') as fh:
            policy = pkl.load(fh)
        taichi_env.apply_agent_action_p(policy.get_actions_p())
        for i in range(self.env.horizon):
            action = policy.get_action_v(i, agent=taichi_env.agent, update=True) if i < self.env.horizon_action else None
            taichi_env.step(action)


def record_target(env, path=None, user_input=False):
    env.reset()
    return Recorder(env).record(user_input)


def replay_policy(env, path=None):
    env.reset()
    Recorder(env).replay_policy(path)
