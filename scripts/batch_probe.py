"""Per-kernel times of fe_step_batch against single-engine stepping on the benchmark block (at rest, restarted every pass)."""
import json
import sys
import time

sys.path.insert(0, '.')
import bench
from fluidlab_amd import _capi

elib = _capi.load_hip()
L = 50
out = {}
for B in (1, 2, 4, 8):
    engs = [bench.build_block(elib, 0, L=L, seed=sd)[0] for sd in range(B)]
    E = type(engs[0])

    def one():
        E.step_batch(engs, 0, 0, L, 0)
        for e in engs:
            e.reset_grad(); e.loss_step_grad(0, L, 0, 1.0, 1.0)
        E.step_grad_batch(engs, 0, 0, L, 0)
    for _ in range(3):
        one()
    for e in engs:
        e.sync()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        one()
    for e in engs:
        e.sync()
    dt = time.perf_counter() - t0
    engs[0].profile_enable(True)
    one(); one()
    prof = engs[0].profile_read()
    engs[0].profile_enable(False)
    out[B] = dict(pairs_per_s_all=round(B * n * L / dt, 1), us_per_pair_per_env=round(1e6 * dt / (n * L * B), 1),
                  kernels_us={k: round(1e3 * v[0] / v[1], 1) for k, v in prof.items() if v[1]})
    print(B, json.dumps(out[B]), flush=True)
    for e in engs:
        e.close()
