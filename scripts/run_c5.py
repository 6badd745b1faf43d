"""BASELINE config 5 as SURVEY 8d C5 writes it, on the GPU: IceCreamDynamic-v0's scene at 256^3 -- a pool of 1,000,000 ICECREAM particles
dispensed by the BallInjector (flux 10 per substep, agent_icecreamdynamic.yaml), the Rigid cone's analytic SDF stand-in, the reference's
40-substep window -- driven by the env's demo policy.  At the reference's fixed dt = 2e-4 the stiff plasto-elastic solid is beyond its
Courant limit on this grid (2.4) and the stream leaves the grid after 490 substeps, 4,900 particles in (profiles/r05_config5_injected_256.txt);
the default here is dt = 5e-5 (40 substeps per step).  Prints the forward rate, the particles in flight and whether the state is finite.
usage: python scripts/run_c5.py [steps] [quality] [n_pool] [dt] [bwd]      (bwd 1: a Solver forward + backward pass over the last `steps`)"""
import contextlib, io, json, sys, time
sys.path.insert(0, '.')
import numpy as np
from fluidlab_amd import _capi
from fluidlab_amd.envs import make

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
quality = float(sys.argv[2]) if len(sys.argv) > 2 else 4
n_pool = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
dt = float(sys.argv[4]) if len(sys.argv) > 4 else 5e-5
elib = _capi.load_hip()
with contextlib.redirect_stdout(io.StringIO()):
    env = make('IceCreamDynamic-v0', seed=0, loss=False, quality=quality, n_pool=n_pool, horizon=steps, inject_till=10**9,
               max_substeps_local=40, ckpt_dest='gpu', engine_lib=elib, dt=dt)
te = env.taichi_env
sim = te.simulator
pol = env.demo_policy()
te.apply_agent_action_p(pol.get_actions_p())
out = {'n_grid': sim.n_grid, 'n_particles': sim.n_particles, 'dt': sim.dt, 'steps': []}
t0 = time.perf_counter(); last = t0
for i in range(steps):
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            te.step(pol.get_action_v(i))
        if (i + 1) % 10 == 0 or i + 1 == steps:
            sim.engine.sync()
            st = sim.engine.get_stats(sim.cur_substep_local)
            x = sim.get_x(sim.cur_substep_local)
            used = int(st['n_used'])
            now = time.perf_counter()
            out['steps'].append({'step': i + 1, 'used': used, 'finite': bool(np.isfinite(x).all()), 'nc': int(st['n_cells_touched']),
                                 'substeps_per_s': round(10 * sim.n_substeps / (now - last), 1)})
            print(out['steps'][-1], flush=True)
            last = time.perf_counter()
    except Exception as e:
        out['error'] = f'step {i}: {str(e)[:160]}'
        print(out['error'], flush=True)
        break
print(json.dumps(out))
