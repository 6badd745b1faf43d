import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import scenarios as S
from fluidlab_amd import _capi
elib = _capi.load_hip()
n, N, mat, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rng = np.random.RandomState(0)
side = (N / 8.0) ** (1 / 3) / n
sc = dict(n_grid=n, N=N, dt=2e-4, gravity=(0.0, -10.0, 0.0), n_substeps=10, boundary=dict(type='cube', lower=(0.05,)*3, upper=(0.95,)*3),
          x=S.f32(rng.uniform(0.3, 0.3 + side, (N, 3))), used=np.ones(N, np.int32), mat=np.full(N, mat, np.int32))
eng = S.make_engine(elib, sc, max_substeps_local=int(sys.argv[5]) if len(sys.argv) > 5 else 12, options={'sort_interval': K})
print('created', flush=True)
NS = int(sys.argv[6]) if len(sys.argv) > 6 else 3
eng.step(0, 0, NS, 0); eng.sync(); print('fwd ok', flush=True)
eng.loss_alloc(1); eng.loss_set_target(0, sc['x']); eng.reset_grad(); eng.loss_step_grad(0, NS, mat, 1.0, 1.0)
eng.step_grad(0, 0, NS, 0); eng.sync(); print('bwd ok', eng.get_stats(1), flush=True)
for it in range(3):
    eng.step(0, 0, NS, 0); eng.reset_grad(); eng.loss_step_grad(0, NS, mat, 1.0, 1.0); eng.step_grad(0, 0, NS, 0); eng.sync(); print('iter', it, 'ok', flush=True)
eng.profile_enable(True); eng.step(0, 0, NS, 0); eng.step_grad(0, 0, NS, 0); print(list(eng.profile_read().items())[:2], flush=True)
print(np.isfinite(eng.get_grad(0)[0]).all())
