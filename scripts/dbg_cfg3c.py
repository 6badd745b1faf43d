"""config-3 debug: per-iteration gradient / action health."""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from fluidlab_amd.envs import make
from fluidlab_amd.optimizer.recorder import Recorder
from fluidlab_amd.optimizer.solver import Solver
from fluidlab_amd.utils.config import load_config

kw = dict(quality=2, particle_density=4e6, n_pool=60000)
env = make('LatteArt-v0', seed=0, loss=False, **kw)
tgt = Recorder(env).record(write=False)
del env
env = make('LatteArt-v0', seed=0, loss=True, target=tgt, **kw)
te = env.taichi_env
eng = te.simulator.engine
cfg = load_config('configs/exp_latteart.yaml').SOLVER
s = Solver(env, None, cfg)
policy = env.trainable_policy(cfg.optim, cfg.init_range)
init = te.get_state()
for it in range(4):
    info, grad = s.forward_backward(init['state'], policy, env.horizon, env.horizon_action)
    g = np.asarray(grad)
    bad = np.argwhere(~np.isfinite(g))
    print(it, 'loss', info['loss'], 'grad finite', np.isfinite(g).all(), 'absmax', np.nanmax(np.abs(g)), 'first bad rows', bad[:3].tolist(), 'n bad', len(bad))
    sl = eng.loss_get(env.horizon)
    badl = np.argwhere(~np.isfinite(sl)).ravel()
    print('   step_loss first nonfinite', badl[:1], 'actions absmax', np.abs(policy.actions_v).max(), np.abs(policy.actions_p).max())
    if len(bad):
        # which frames' particle adjoints are non-finite (only the last chunk's are still resident)
        for f in (3299, 3000, 2000, 1000, 100, 1, 0):
            pass
    policy.optimize(grad, info)
