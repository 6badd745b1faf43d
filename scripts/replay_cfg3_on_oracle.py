"""config-3 cross-check: replay the iteration-2 actions (the iterate whose rollout goes non-finite at 128^3) forward on
the CPU oracle (fp32, OpenMP) and on the HIP engine; report the first non-finite step of each."""
import os, sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from fluidlab_amd import _capi
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
cfg = load_config('configs/exp_latteart.yaml').SOLVER
s = Solver(env, None, cfg)
policy = env.trainable_policy(cfg.optim, cfg.init_range)
init = te.get_state()
for it in range(2):
    info, grad = s.forward_backward(init['state'], policy, env.horizon, env.horizon_action)
    policy.optimize(grad, info)
acts_v, acts_p = policy.actions_v.copy(), policy.actions_p.copy()
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 135
LWIN = int(sys.argv[2]) if len(sys.argv) > 2 else 50      # 50 = the reference's window; >= 10*NS reproduces the resident-trajectory noise
del env, te, s


def replay(lib, label):
    """same scene with the reference's 50-substep window (latteart_env.py:31) so that both engines see the same
    locally-random injection noise"""
    env = make('LatteArt-v0', seed=0, loss=False, engine_lib=lib, max_substeps_local=LWIN, **kw)
    te = env.taichi_env
    te.set_state(init['state'], grad_enabled=False)
    te.apply_agent_action_p(acts_p)
    t0 = time.time()
    snaps, first = {}, None
    for i in range(NS):
        te.step(acts_v[i])
        if i % 10 == 9 or i >= NS - 12:
            try:
                st = te.get_state()['state']
            except _capi.FeEngineError as e:
                print(label, 'engine error at step', i, ':', e)
                first = i
                break
            ok = np.isfinite(st['x'][st['used'] > 0]).all()
            if not ok:
                first = i
                break
            snaps[i] = (st['x'].copy(), st['used'].copy())
    print(label, 'first non-finite step (10-step granularity)', first, 'in %.1fs' % (time.time() - t0), flush=True)
    return snaps, first

a, fa = replay(None, 'hip')
os.environ.setdefault('OMP_NUM_THREADS', '32')
b, fb = replay(_capi.EngineLib('oracle/_build/libfe_oracle_f32.so'), 'oracle-f32')
for i in sorted(set(a) & set(b)):
    xa, ua = a[i]; xb, ub = b[i]
    m = ua > 0
    d = np.linalg.norm(xa[m] - xb[m], axis=1)
    print('step', i, 'used equal', bool((ua == ub).all()), 'n_used', int(m.sum()), 'x relL2', float(np.linalg.norm(xa[m] - xb[m]) / np.linalg.norm(xb[m])),
          'max|dx|/dx', float(d.max() * 128))
