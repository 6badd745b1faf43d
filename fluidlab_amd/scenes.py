"""Benchmark / smoke scenes of the package and the helper that turns a scene dict into an engine through the C ABI.

A scene is a plain dict of float32-representable numpy arrays (so the fp64 oracle and the fp32 HIP engine see bit-identical
inputs).  `water_block` is BASELINE config 2; bench.py, scripts/ and tests/scenarios.py use it from here.
"""
import numpy as np

from fluidlab_amd._capi import Engine

WATER, MILK, COFFEE, ELASTIC, ICECREAM, RIGID, RIGID_HEAVY, MILK_VIS = 0, 1, 2, 3, 4, 5, 6, 8
MAT_LIQUID, MAT_PLASTO_ELASTIC, MAT_ELASTIC, MAT_RIGID = 200, 201, 202, 203
# (mu, lam, rho, class) -- fluidlab/configs/macros.py:65-83,143-201
MATERIALS = {
    WATER: (0.0, 277.78, 1.0, MAT_LIQUID), MILK: (0.0, 277.78, 0.5, MAT_LIQUID), COFFEE: (0.0, 277.78, 1.0, MAT_LIQUID),
    ELASTIC: (416.67, 277.78, 1.0, MAT_ELASTIC), ICECREAM: (416.67, 277.78, 0.5, MAT_PLASTO_ELASTIC),
    MILK_VIS: (200.0, 277.78, 1.0, MAT_LIQUID),
    RIGID: (416.67, 277.78, 1.0, MAT_RIGID), RIGID_HEAVY: (416.67, 277.78, 10.0, MAT_RIGID),
}


def f32(a):
    return np.asarray(a, dtype=np.float32)


def water_block(n_grid=32, n_particles=4096, seed=0, lo=0.30, hi=0.53, gravity=(0.0, -10.0, 0.0), mat=WATER):
    """BASELINE config 2: one block of `mat` at rest, x ~ U([lo, hi]^3), in a cube boundary."""
    rng = np.random.RandomState(seed)
    N = n_particles
    return dict(
        n_grid=n_grid, N=N, dt=2e-4, gravity=gravity, n_substeps=10,
        boundary=dict(type='cube', lower=(0.05, 0.05, 0.05), upper=(0.95, 0.95, 0.95)),
        x=f32(rng.uniform(lo, hi, (N, 3))), used=np.ones(N, np.int32), mat=np.full(N, mat, np.int32),
    )


def make_engine(elib, sc, max_substeps_local=None, device=0, options=None):
    n = sc['n_grid']
    L = max_substeps_local or sc.get('max_substeps_local', 64)
    eng = Engine(elib, n_grid=n, n_particles=sc['N'], max_substeps_local=L, n_substeps=sc['n_substeps'],
                 max_action_steps=sc.get('horizon', 8), dt=sc['dt'], p_vol=(0.5 / n) ** 2, gravity=sc['gravity'],
                 boundary=elib.make_boundary(**sc['boundary']), device=device)
    for k, v in (options or sc.get('options') or {}).items():
        eng.set_option(k, v)
    for st in sc.get('statics', ()):
        eng.add_static(st['voxels'], st['T'], friction=st['friction'])
    mat = sc['mat']
    props = np.array([MATERIALS[int(m)] for m in mat], dtype=np.float64)
    eng.init_particles(sc['x'], sc['used'], mat, props[:, 3].astype(np.int32), props[:, 0], props[:, 1], props[:, 2],
                       np.asarray(sc.get('body_id', np.zeros(sc['N'], np.int32)), np.int32))
    if any(k in sc for k in ('v', 'C', 'F')):
        eng.set_frame(0, v=sc.get('v'), C_=sc.get('C'), F=sc.get('F'))
    return eng


def get_state(eng, f):
    N, dt = eng.N, eng.dtype
    x = np.zeros((N, 3), dt); v = np.zeros((N, 3), dt); C = np.zeros((N, 3, 3), dt); F = np.zeros((N, 3, 3), dt)
    used = np.zeros((N,), np.int32)
    eng.get_frame(f, x, v, C, F, used)
    return dict(x=x, v=v, C=C, F=F, used=used)
