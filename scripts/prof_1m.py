"""The 256^3 / 1M-particle block of bench.py's `extra.config5_water_256_1M` on its own, for rocprofv3 (kernel trace / --pmc passes):
bench.extra_block's scene, one untimed window of 10 pairs, then REPS windows.  usage: prof_1m.py [reps] [water|icecream]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from fluidlab_amd import _capi, scenes as S

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mat = S.ICECREAM if (len(sys.argv) > 2 and sys.argv[2] == 'icecream') else S.WATER
n_grid, n, L = 256, 1_000_000, 10
elib = _capi.load_hip()
rng = np.random.RandomState(0)
side = (n / 8.0) ** (1 / 3) / n_grid
sc = S.water_block(n_grid=n_grid, n_particles=n, seed=0, mat=mat)
sc['x'] = S.f32(rng.uniform(0.3, 0.3 + side, (n, 3)))
eng = S.make_engine(elib, sc, max_substeps_local=L, device=0)
eng.loss_alloc(1); eng.loss_set_target(0, sc['x'])
for _ in range(1 + reps):
    bench.window_step(eng, L, mat=mat, roll=False)
eng.sync()
st = eng.get_stats(L // 2)
print('n_used', int(st['n_used']), 'n_cells_touched', int(st['n_cells_touched']), 'pairs', (1 + reps) * L)
eng.close()
