"""BASELINE config 5's scene as the reference defines it -- IceCreamDynamic-v0: 64^3 grid, 100k-particle ICECREAM pool
dispensed by a BallInjector (flux 10 per substep until substep 7700), a Rigid cone with an SDF mesh, 900 steps x 10
substeps -- one full Solver iteration (forward with loss + backward) on one MI355X, whole trajectory resident in HBM."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from fluidlab_amd.envs import make
from fluidlab_amd.optimizer.recorder import Recorder
from fluidlab_amd.optimizer.solver import Solver
from fluidlab_amd.utils.config import load_config

kw = dict(max_substeps_local=None)
t0 = time.time()
env = make('IceCreamDynamic-v0', seed=0, loss=False, **kw)
tgt = Recorder(env).record(write=False)
t_rec = time.time() - t0
n_used = int(tgt['used'][-1].sum())
x_end = tgt['x'][-1][tgt['used'][-1] > 0]
del env
env = make('IceCreamDynamic-v0', seed=0, loss=True, target=tgt, **kw)
eng = env.taichi_env.simulator.engine
cfg = load_config('configs/exp_icecream_dynamic.yaml').SOLVER
pol = env.trainable_policy(cfg.optim, cfg.init_range)
demo = env.demo_policy()
pol.actions_v[:] = demo.actions_v; pol.actions_p[:] = demo.actions_p
pol.actions_v[200:, 0] += 0.0002
env.taichi_env.loss.temporal_range[1] = env.horizon
s = Solver(env, None, cfg)
infos = []
for it in range(3):
    info, g = s.forward_backward(env.taichi_env.get_state()['state'], pol, env.horizon, env.horizon_action)
    infos.append(dict(loss=float(info['loss']), fwd=info['forward_s'], bwd=info['backward_s'], grad_finite=bool(np.isfinite(g).all()),
                      grad_absmax=float(np.nanmax(np.abs(g)))))
st = eng.get_stats(8999)
sub = 9000
out = dict(n_particles=int(env.taichi_env.n_particles), n_used_end=n_used, finite_end=bool(np.isfinite(x_end).all()),
           y_min_end=float(x_end[:, 1].min()), record_s=round(t_rec, 2), iters=infos,
           fwd_substeps_per_s=round(sub / infos[-1]['fwd'], 1), pairs_per_s=round(sub / (infos[-1]['fwd'] + infos[-1]['bwd']), 1),
           bytes_state_GB=round(st['bytes_state'] / 2**30, 2))
print(json.dumps(out))
json.dump(out, open('gpurun_out/icecream_dynamic.json', 'w'), indent=1)
