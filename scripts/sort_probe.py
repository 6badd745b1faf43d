"""Stage times of the sort (option prof_fine) for a block of n particles on an n_grid^3 grid."""
import sys, json
sys.path.insert(0, '.')
import numpy as np
import bench
from fluidlab_amd import scenes as S
from fluidlab_amd._capi import load_hip
n_grid, n = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.RandomState(0)
side = (n / 8.0) ** (1 / 3) / n_grid
sc = S.water_block(n_grid=n_grid, n_particles=n, seed=0)
sc['x'] = S.f32(rng.uniform(0.3, 0.3 + side, (n, 3)))
eng = S.make_engine(load_hip(), sc, max_substeps_local=40)
eng.loss_alloc(1); eng.loss_set_target(0, sc['x'])
bench.window_step(eng, 40, roll=False); eng.sync()
eng.set_option('prof_fine', 1)
eng.profile_enable(True); bench.window_step(eng, 40, roll=False); prof = eng.profile_read(); eng.profile_enable(False)
print(json.dumps({k: [round(1e3 * v[0] / v[1], 1), v[1]] for k, v in prof.items() if v[1]}))
