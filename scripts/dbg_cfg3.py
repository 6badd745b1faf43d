import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from fluidlab_amd.envs import make
q = float(sys.argv[1]) if len(sys.argv) > 1 else 2
dens = float(sys.argv[2]) if len(sys.argv) > 2 else 1.73e6
env = make('LatteArt-v0', seed=0, loss=False, quality=q, particle_density=dens, n_pool=104000)
te = env.taichi_env; sim = te.simulator
pol = env.demo_policy()
te.set_state(**te.get_state())
te.apply_agent_action_p(pol.get_actions_p())
for i in range(40):
    te.step(pol.get_action_v(i))
    if i % 4 == 0 or i < 3:
        st = sim.get_state()
        u = st['used'].astype(bool)
        x = st['x'][u]
        print(i, 'used', u.sum(), 'finite', np.isfinite(x).all(), 'x range', np.nanmin(x, 0), np.nanmax(x, 0), 'vmax', np.nanmax(np.abs(st['v'][u])), 'Fdet min', np.nanmin(np.linalg.det(st['F'][u])), flush=True)
        if not np.isfinite(x).all(): break
