"""B environments of one process stepped in lockstep (BASELINE config 4 with B envs per GPU).

The reference steps one environment per process (solver.py:17-72).  Here the B replicas of a rank -- same scene, own injector noise --
share every launch of a substep: `MPMSimulator.step_` is cut into begin / engine call / end (mpm:735-753), the B engine calls become ONE
`fe_step_batch` crossing (gridDim.y = B, csrc/fe_engine.hip "Batched environments"), and likewise backwards.  Everything else -- action
buffering, losses, checkpoints, the agents' action adjoints -- stays per environment, in the order `TaichiEnv.step` / `step_grad` has it.
"""
from time import time

import numpy as np


class EnvBatch:
    def __init__(self, envs):
        assert len(envs) >= 1
        self.envs = list(envs)
        self.tes = [e.taichi_env for e in self.envs]
        self.sims = [te.simulator for te in self.tes]
        self.engines = [s.engine for s in self.sims]
        self.Engine = type(self.engines[0])
        for s in self.sims:
            assert s.smoke_field is None or len(self.envs) == 1, 'smoke fields step per environment: batch of one only'

    def _same_args(self, actions):
        args = {s.step_args(a) for s, a in zip(self.sims, actions)}
        assert len(args) == 1, f'the environments of a batch have to be at the same substep with the same kind of action: {args}'
        return args.pop()

    def step(self, actions):
        """TaichiEnv.step for every environment, the engine part once (taichi_env.py:155-160, mpm:721-753)."""
        actions = [te._as_action(a) for te, a in zip(self.tes, actions)]
        for s, a in zip(self.sims, actions):
            if s.grad_enabled and s.cur_substep_local == 0:
                s.actions_buffer = []
            s.step_begin(a)
        self.Engine.step_batch(self.engines, *self._same_args(actions))
        for te, s, a in zip(self.tes, self.sims, actions):
            s.step_end(a)
            if s.grad_enabled:
                s.actions_buffer.append(a)
            if s.cur_substep_local == 0:
                s.memory_to_cache()
            if te.loss:
                te.loss.step()
            te.t += 1

    def step_grad(self, actions):
        """TaichiEnv.step_grad for every environment (taichi_env.py:162-166, mpm:755-775)."""
        actions = [te._as_action(a) for te, a in zip(self.tes, actions)]
        for te, s in zip(self.tes, self.sims):
            if te.loss:                                 # the loss of a step is differentiated before the step itself
                te.loss.step_grad()
            s.step_grad_begin()
        self.Engine.step_grad_batch(self.engines, *self._same_args(actions))
        for s, a in zip(self.sims, actions):
            s.step_grad_end(a)

    def forward_backward(self, sim_states, policies, horizon, horizon_action):
        """Solver.forward_backward (solver.py:23-59) for the whole batch: [(loss_info, dLoss/d comp_actions)] per environment.
        `policies`: one per environment (the same object for replicas of a plain actions table; staged policies need one object each)."""
        tes = self.tes
        # A staged policy (StagePlan with a RETURN stage) writes its scripted actions into its own table from the environment it is asked
        # for: shared between replicas that diverge (injector noise), every replica would read back the LAST one's action in the reverse
        # sweep.  Replicas of a plain actions table may share the object; staged ones need a copy each.
        for a in range(len(policies)):
            for b in range(a):
                assert policies[a] is not policies[b] or getattr(policies[a], 'plan', None) is None, \
                    'EnvBatch.forward_backward: environments must not share a staged policy object (copy it per environment)'
        for te, st in zip(tes, sim_states):
            te.set_state(st, grad_enabled=True)
        t1 = time()
        for te, pol in zip(tes, policies):
            te.apply_agent_action_p(pol.get_actions_p())
        cur_horizon = tes[0].loss.temporal_range[1]
        assert all(te.loss.temporal_range[1] == cur_horizon for te in tes)
        for i in range(cur_horizon):
            self.step([pol.get_action_v(i, agent=te.agent, update=True) if i < horizon_action else None for te, pol in zip(tes, policies)])
        infos = [te.get_final_loss() for te in tes]
        t2 = time()
        for te in tes:
            te.reset_grad()
            te.get_final_loss_grad()
        freeze_till = policies[0].freeze_till
        assert all(p.freeze_till == freeze_till for p in policies)
        for i in range(cur_horizon - 1, freeze_till - 1, -1):
            self.step_grad([pol.get_action_v(i) if i < horizon_action else None for pol in policies])
        grads = []
        for te, pol in zip(tes, policies):
            te.apply_agent_action_p_grad(pol.get_actions_p())
            grads.append(np.asarray(te.agent.get_grad(horizon_action)))
        t3 = time()
        for info in infos:
            info['forward_s'], info['backward_s'] = t2 - t1, t3 - t2
        return list(zip(infos, grads))
