import sys, time, json
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from fluidlab_amd.envs import make
from fluidlab_amd.optimizer.recorder import Recorder
from fluidlab_amd.optimizer.solver import Solver
from fluidlab_amd.utils.config import load_config
H = int(sys.argv[1]) if len(sys.argv) > 1 else 300
kw = dict(max_substeps_local=None, horizon=H, inject_till=2500)
env = make('IceCreamDynamic-v0', seed=0, loss=False, **kw)
tgt = Recorder(env).record(write=False)
del env
env = make('IceCreamDynamic-v0', seed=0, loss=True, target=tgt, **kw)
eng = env.taichi_env.simulator.engine
cfg = load_config('configs/exp_icecream_dynamic.yaml').SOLVER
pol = env.trainable_policy(cfg.optim, cfg.init_range)
demo = env.demo_policy()
pol.actions_v[:] = demo.actions_v; pol.actions_p[:] = demo.actions_p
env.taichi_env.loss.temporal_range[1] = env.horizon
s = Solver(env, None, cfg)
s.forward_backward(env.taichi_env.get_state()['state'], pol, env.horizon, env.horizon_action)
eng.set_option("prof_fine", 1)
eng.profile_enable(True)
info, g = s.forward_backward(env.taichi_env.get_state()['state'], pol, env.horizon, env.horizon_action)
prof = eng.profile_read(); eng.profile_enable(False)
print('fwd', info['forward_s'], 'bwd', info['backward_s'])
print({k: (round(1e3 * v[0] / v[1], 1), v[1]) for k, v in prof.items() if v[1]})
print(eng.get_stats(H * 10 - 1))
