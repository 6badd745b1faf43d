"""Circulation-v0's smoke solver at the reference size (128^3 smoke grid, free slab 60 < j < 68, room SDF) on one MI355X:
time per smoke step forward / backward for 50 (reference env) and 500 (SmokeField default) Jacobi sweeps."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from fluidlab_amd.envs import make

out = {}
for iters in (50, 500):
    env = make('Circulation-v0', seed=0, loss=False, res=128, horizon=40, solver_iters=iters, max_substeps_local=None)
    te = env.taichi_env
    eng, sf, sim = te.simulator.engine, te.smoke_field, te.simulator
    e = 0
    eng.eff_set_action(e, 0, 0, 10, np.array([0.0, 0.0, 0.0, 0.0, 0.1, 0.0, 0.02, 0.04]))
    n_free = int((sf.get_state(0)['q'][..., 0] > 0).sum())
    S = 30
    for s in range(3):
        eng.smoke_step(s, 10 * s)
    eng.sync()
    t0 = time.time()
    for s in range(3, S):
        eng.eff_set_action(e, s, s, 10, np.array([0.0, 0.0, 0.0, 0.0, 0.1, 0.0, 0.02, 0.04]))
        eng.smoke_step(s, 10 * s)
    eng.sync()
    fwd = (time.time() - t0) / (S - 3)
    eng.reset_grad()
    gq = np.zeros((128, 128, 128, 1), np.float32); gq[20:100, 64, 20:100] = 1.0
    eng.smoke_add_grad(S, gq=gq)
    eng.sync()
    t0 = time.time()
    for s in reversed(range(3, S)):
        eng.smoke_step_grad(s, 10 * s)
    eng.sync()
    bwd = (time.time() - t0) / (S - 3)
    g = eng.eff_get_action_grad(e, 0, S, 8)
    slab_cells = 128 * 7 * 128
    # algorithmic bytes of one Jacobi sweep over the slab: read p (6 neighbours cached -> 1), div, mask; write p'
    sweep_bytes = slab_cells * (4 + 4 + 1 + 4)
    out[f'iters{iters}'] = dict(fwd_ms=round(1e3 * fwd, 3), bwd_ms=round(1e3 * bwd, 3), us_per_sweep_fwd=round(1e6 * fwd / iters, 2),
                                slab_cells=slab_cells, sweep_GBps_fwd=round(sweep_bytes * iters / fwd / 1e9, 1),
                                action_grad_finite=bool(np.isfinite(g).all()), action_grad_absmax=float(np.abs(g).max()))
    print(json.dumps(out[f'iters{iters}']))
    del env
json.dump(out, open('gpurun_out/smoke_128.json', 'w'), indent=1)
