"""Per-kernel times and work list of the LatteArt-v0 replica bench.py --gpus N runs (config 3 scene), forward + backward of a few steps."""
import json, sys, time
sys.path.insert(0, '.')
import numpy as np
import bench
from fluidlab_amd import _capi
from fluidlab_amd.envs import make
elib = _capi.load_hip()
scene = sys.argv[1] if len(sys.argv) > 1 else 'config3'
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
env = make('LatteArt-v0', seed=0, loss=False, engine_lib=elib, **bench.C4_SCENES[scene])
te = env.taichi_env
sim = te.simulator
eng = sim.engine
pol = env.demo_policy()
te.apply_agent_action_p(pol.get_actions_p())
for phase in range(3):
    eng.profile_enable(True)
    eng.sync(); t0 = time.perf_counter()
    for i in range(n_steps):
        te.step(pol.get_action_v(phase * n_steps + i))
    t_enq = time.perf_counter() - t0
    eng.sync(); dt = time.perf_counter() - t0
    prof = eng.profile_read(); eng.profile_enable(False)
    f = sim.cur_substep_local
    st = eng.get_stats(f); ws = eng.get_work_stats(f)
    print(json.dumps({'steps': [phase * n_steps, (phase + 1) * n_steps], 'fwd_substeps_per_s': round(n_steps * sim.n_substeps / dt, 1), 'host_enqueue_us_per_substep': round(1e6 * t_enq / (n_steps * sim.n_substeps), 1), 'total_us_per_substep': round(1e6 * dt / (n_steps * sim.n_substeps), 1), 'n_used': st['n_used'], 'nc': st['n_cells_touched'], 'slow': st['n_slow_path'],
                      **ws, 'us': {k: round(1e3 * v[0] / v[1], 1) for k, v in prof.items() if v[1]}}))
