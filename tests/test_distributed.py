"""Env-level data parallelism on CPU: world_size 2 and 8 (the driver's node: one rank per GPU), gloo.  Each rank
owns one environment replica with its own injector randomness; the action gradients are all-reduced once per optimisation pass
and the replicated Adam state must stay bit-identical on every rank."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, pickle
import numpy as np
sys.path.insert(0, %(root)r)
from fluidlab_amd._capi import EngineLib
from fluidlab_amd.envs import make
from fluidlab_amd.optimizer.distributed import EnvParallel
from fluidlab_amd.optimizer.recorder import Recorder
from fluidlab_amd.optimizer.solver import Solver
from fluidlab_amd.utils.config import load_config
par = EnvParallel(backend='gloo')
lib = EngineLib(os.path.join(%(root)r, 'oracle', '_build', 'libfe_oracle_f32.so'))
kw = dict(quality=0.5, particle_density=4e4, n_pool=300, horizon=6, horizon_action=4, engine_lib=lib)
tgt = Recorder(make('LatteArt-v0', seed=0, loss=False, **kw)).record(write=False)
env = make('LatteArt-v0', seed=100 + par.rank, loss=True, target=tgt, **kw)      # per-rank injector randomness
cfg = load_config('configs/exp_latteart.yaml').SOLVER
cfg.n_iters = 2
np.random.seed(0)
log = []
solver = Solver(env, None, cfg, parallel=par)
policy = env.trainable_policy(cfg.optim, cfg.init_range)
init = env.taichi_env.get_state()
for it in range(2):
    info, g_local = solver.forward_backward(init['state'], policy, env.horizon, env.horizon_action)
    g_mean, _ = par.all_reduce_mean(g_local, [info['loss']])
    policy.optimize(g_mean, info)
    log.append(dict(g_local=g_local, g_mean=g_mean, actions=policy.comp_actions.copy(), loss=info['loss']))
pickle.dump(log, open(os.path.join(%(out)r, f'rank{par.rank}.pkl'), 'wb'))
par.close()
'''


@pytest.mark.parametrize('world', [2, 8])
def test_gloo_action_gradient_allreduce(tmp_path, world):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % dict(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='2' if world == 2 else '1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(29517 + world), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    import pickle
    logs = [pickle.load(open(tmp_path / f'rank{k}.pkl', 'rb')) for k in range(world)]
    for it in range(2):
        rows = [lg[it] for lg in logs]
        for k in range(1, world):
            assert not np.array_equal(rows[0]['g_local'], rows[k]['g_local'])      # replicas really differ (per-rank injector noise)
        mean = np.mean([r_['g_local'].astype(np.float32) for r_ in rows], axis=0)
        assert np.allclose(rows[0]['g_mean'], mean, rtol=2e-6, atol=1e-12)
        for k in range(1, world):
            assert np.array_equal(rows[0]['g_mean'], rows[k]['g_mean'])            # identical on every rank ...
            assert np.array_equal(rows[0]['actions'], rows[k]['actions'])          # ... so the policies stay bit-identical


def test_pin_to_cores_gives_every_rank_of_a_node_its_own_cores():
    """bench.py pins each rank to an equal, contiguous share of the cores (the host enqueues most of a core's worth of launches per
    rank): eight ranks of one node get eight disjoint, non-empty sets that cover what the process may run on."""
    sys.path.insert(0, ROOT)
    import bench
    for cpus in (list(range(256)), list(range(8)), [3, 5, 9, 11, 12, 13, 20, 21, 22, 40], list(range(5))):
        sets = [bench.core_share(r, 8, cpus) for r in range(8)]
        assert all(len(s_) >= 1 for s_ in sets)
        if len(cpus) >= 8:
            flat = [c for s_ in sets for c in s_]
            assert len(flat) == len(set(flat)) and set(flat) <= set(cpus)            # disjoint
            assert max(len(s_) for s_ in sets) - min(len(s_) for s_ in sets) <= 0   # equal shares
    assert bench.core_share(0, 1, [0, 1, 2, 3]) == [0, 1, 2, 3]
