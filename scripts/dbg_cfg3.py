import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from fluidlab_amd.envs import make
q = float(sys.argv[1]) if len(sys.argv) > 1 else 2
dens = float(sys.argv[2]) if len(sys.argv) > 2 else 1.73e6
ZERO = int(sys.argv[3]) if len(sys.argv) > 3 else 0
env = make('LatteArt-v0', seed=0, loss=False, quality=q, particle_density=dens, n_pool=60000)
te = env.taichi_env; sim = te.simulator
pol = env.demo_policy()
te.set_state(**te.get_state())
te.apply_agent_action_p(pol.get_actions_p() if ZERO == 0 else np.array([0.15, 0.65, 0.5]))
for i in range(330):
    te.step((pol.get_action_v(i) if ZERO == 0 else np.zeros(3)) if i < 250 else None)
    if i % 30 == 0 or i == 329:
        st = sim.get_state()
        u = st['used'].astype(bool)
        x = st['x'][u]
        print(i, 'used', u.sum(), 'finite', np.isfinite(x).all(), 'nan rows', int((~np.isfinite(x).all(1)).sum()), 'y range', float(np.nanmin(x[:, 1])), float(np.nanmax(x[:, 1])), 'vmax', np.nanmax(np.abs(st['v'][u])), 'Fdet min', np.nanmin(np.linalg.det(st['F'][u])), flush=True)
