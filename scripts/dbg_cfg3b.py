import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from fluidlab_amd.envs import make
env = make('LatteArt-v0', seed=0, loss=False, quality=2, particle_density=4e6, n_pool=60000)
te = env.taichi_env; sim = te.simulator; eng = sim.engine
te.set_state(**te.get_state())
te.apply_agent_action_p(np.array([0.15, 0.65, 0.5]))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
eng.set_option('sort_interval', K)
a = np.zeros(3, np.float32) if len(sys.argv) < 3 else np.array([float(sys.argv[2]), 0, 0], np.float32)
sim.agent.set_action(s=0, s_global=0, n_substeps=10, action=a)
for f in range(10):
    eng.substep(f, f, True)
    st = sim.engine
    x = np.zeros((sim.n_particles, 3), np.float32); v = np.zeros_like(x); u = np.zeros(sim.n_particles, np.int32)
    eng.get_frame(f + 1, x=x, v=v, used=u)
    ub = u.astype(bool)
    print(f, 'used', ub.sum(), 'nan rows', int((~np.isfinite(x[ub]).all(1)).sum()), 'vmax', float(np.nanmax(np.abs(v[ub]))), 'eff', sim.agent.get_state(f + 1)[0][:3], eng.get_stats(f + 1), flush=True)
