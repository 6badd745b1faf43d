"""Host time inside fe_step / fe_step_grad as a function of how many substeps one call enqueues, on an idle queue (benchmark block): does the call return before the GPU is done
(enqueue cost), or does it wait for it (back-pressure of the queue)?"""
import sys, time
sys.path.insert(0, '.')
import bench
from fluidlab_amd._capi import load_hip
eng, sc = bench.build_block(load_hip(), 0)
L = bench.CHUNK
for _ in range(3): bench.window_step(eng, L)
eng.sync()
for n in (1, 2, 5, 10, 20, 50, 100):
    tf = tb = df = db = 0.0
    R = 5
    for _ in range(R):
        eng.sync(); t0 = time.perf_counter(); eng.step(0, 0, n, 0); t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter()
        tf += t1 - t0; df += t2 - t0
        eng.reset_grad(); eng.loss_step_grad(0, n, 0, 1.0, 1.0)
        eng.sync(); t0 = time.perf_counter(); eng.step_grad(0, 0, n, 0); t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter()
        tb += t1 - t0; db += t2 - t0
    print(f'n = {n:3d} substeps per call:  fe_step returns after {1e6 * tf / R / n:6.1f} us per substep (GPU done after {1e6 * df / R / n:6.1f});  fe_step_grad {1e6 * tb / R / n:6.1f} ({1e6 * db / R / n:6.1f})')
