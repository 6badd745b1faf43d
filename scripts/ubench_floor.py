"""Where does the time of a nearly empty substep go?  A small block of used water plus an unused pool of varying size."""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import scenarios as S
from fluidlab_amd._capi import load_hip

lib = load_hip()
for n_used, n_pool, n_grid in ((2000, 0, 64), (2000, 200000, 64), (200000, 0, 128), (2000, 0, 128)):
    sc = S.water_block(n_grid=n_grid, n_particles=n_used)
    if n_pool:
        N = n_used + n_pool
        sc['x'] = np.concatenate([np.tile(S.f32([-100, -100, -100]), (n_pool, 1)), sc['x']])
        sc['v'] = np.concatenate([np.zeros((n_pool, 3), np.float32), sc['v']]) if 'v' in sc else None
        sc['used'] = np.concatenate([np.zeros(n_pool, np.int32), sc['used']])
        sc['mat'] = np.concatenate([np.full(n_pool, sc['mat'][0], np.int32), sc['mat']])
        sc['N'] = N
        if sc.get('v') is None: sc.pop('v', None)
    L = 40
    eng = S.make_engine(lib, sc, max_substeps_local=L + 1)
    for rep in range(2):
        if rep == 1: eng.profile_enable(True)
        for f in range(L): eng.substep(f, f, 0)
        eng.reset_grad()
        for f in reversed(range(L)): eng.substep_grad(f, f, 0)
        eng.sync()
    prof = eng.profile_read(); eng.profile_enable(False)
    print(n_used, n_pool, n_grid, {k: round(v[0] * 1e3 / max(v[1], 1), 1) for k, v in prof.items() if v[1]}, flush=True)
    eng.close()
