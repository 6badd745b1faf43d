"""The four remaining task environments at the reference's sizes (64^3 grid, particle_density 1e6), one Solver iteration each
(forward with loss + backward) on one MI355X with the trajectory resident in HBM; the horizon is shortened to keep the run to
seconds (throughput does not depend on it).  Usage: python scripts/run_envs.py [horizon]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from fluidlab_amd.envs import make
from fluidlab_amd.optimizer.solver import Solver
from fluidlab_amd.utils.config import load_config

H = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ONLY = sys.argv[2].split(',') if len(sys.argv) > 2 else None


def pouring(pol):
    pol.actions_v[:, 5] = 0.004


def transporting(pol):
    pol.actions_p[:] = [0.42, 0.5, 0.5, 0.0, 0.0, 0.0]
    pol.actions_v[:, 5] = 0.0005


def mixing(pol):
    pol.actions_p[:] = [0.5, 0.62, 0.5]
    pol.actions_v[:, 0] = 0.003


def gathering_o(pol):
    pol.actions_v[:, 0] = 0.003


out = {}
for name, cfg_file, prepare in (('Pouring-v0', 'configs/exp_pouring.yaml', pouring), ('Transporting-v0', 'configs/exp_transporting.yaml', transporting),
                                ('Mixing-v0', 'configs/exp_mixing.yaml', mixing), ('GatheringO-v0', 'configs/exp_gatheringO.yaml', gathering_o)):
    if ONLY and name not in ONLY:
        continue
    t0 = time.time()
    env = make(name, seed=0, loss=True, horizon=H, max_substeps_local=None)
    t_build = time.time() - t0
    te = env.taichi_env
    if hasattr(te.loss, 'temporal_range'):
        te.loss.temporal_range[1] = env.horizon
    cfg = load_config(cfg_file).SOLVER
    pol = env.trainable_policy(cfg.optim, cfg.init_range)
    prepare(pol)
    s = Solver(env, None, cfg)
    its = []
    for it in range(2):
        info, g = s.forward_backward(te.get_state()['state'], pol, env.horizon, env.horizon_action)
        its.append(dict(loss=float(info['loss']), fwd=round(info['forward_s'], 3), bwd=round(info['backward_s'], 3), grad_finite=bool(np.isfinite(g).all()),
                        grad_absmax=float(np.nanmax(np.abs(g)))))
    sub = H * te.simulator.n_substeps
    eng = te.simulator.engine
    eng.profile_enable(True)                                  # per-kernel HIP-event times of one more iteration
    s.forward_backward(te.get_state()['state'], pol, env.horizon, env.horizon_action)
    prof = eng.profile_read(); eng.profile_enable(False)
    kern = {k: round(v[0] * 1e3 / max(v[1], 1), 1) for k, v in prof.items() if v[1]}
    x = np.zeros((te.n_particles, 3), eng.dtype); used = np.zeros((te.n_particles,), np.int32)
    eng.get_frame(sub, x=x, used=used)
    st = eng.get_stats(sub - 1)
    out[name] = dict(n_particles=int(te.n_particles), n_used_end=int(used.sum()), finite_end=bool(np.isfinite(x[used > 0]).all()), horizon=H,
                     build_s=round(t_build, 2), iters=its, fwd_substeps_per_s=round(sub / its[-1]['fwd'], 1),
                     pairs_per_s=round(sub / (its[-1]['fwd'] + its[-1]['bwd']), 1), slow_path=int(st['n_slow_path']), kernel_us=kern,
                     bytes_state_GB=round(st['bytes_state'] / 2**30, 2))
    print(name, json.dumps(out[name]), flush=True)
    del s, env, te, eng
json.dump(out, open('gpurun_out/envs_full_size.json', 'w'), indent=1)
