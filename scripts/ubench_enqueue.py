"""How long does the host take to enqueue a window of substeps, against how long the GPU takes to run it?  (evolving block of bench.py)"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench
from fluidlab_amd._capi import load_hip
eng, sc = bench.build_block(load_hip(), 0)
for _ in range(3): bench.window_step(eng, bench.CHUNK)
eng.sync()
for back in (False, True):
    te = tg = 0.0
    for _ in range(10):
        eng.sync(); t0 = time.perf_counter()
        bench.window_step(eng, bench.CHUNK, backward=back)
        t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter()
        te += t1 - t0; tg += t2 - t0
    n = 10 * bench.CHUNK
    print('fwd+bwd' if back else 'fwd only', 'host enqueue us per substep(pair):', round(1e6 * te / n, 1), ' until done:', round(1e6 * tg / n, 1))
