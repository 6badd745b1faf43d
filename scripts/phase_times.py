"""Debug: wall-clock phases of k_p2g per workgroup (option dbg=8, s_memrealtime @100 MHz)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench
from fluidlab_amd import _capi
elib = _capi.load_hip()
eng, sc = bench.build_engine(elib, 0)
for _ in range(2):
    bench.one_step(eng, 20, backward=False)
eng.set_option('dbg', 8)
eng.sync()
eng.step(0, 0, 1, 0)        # one substep (sorts first)
eng.step(1, 1, 1, 0)        # a substep without a sort: its p2g overwrites the stamps
eng.sync()
ts = np.zeros(8 * 4096, np.uint64)
elib.lib.fe_debug_timestamps(eng.h, ts.ctypes.data_as(C.c_void_p), ts.size)
ts = ts.reshape(4096, 8).astype(np.int64)
act = ts[:, 4] > 0
t = ts[act]
t0 = t[:, 0].min()
print('workgroups with an item:', act.sum(), ' launched stamps:', (ts[:, 0] > 0).sum())
print('kernel span (first start -> last flush end): %.2f us' % ((t[:, 4].max() - t0) / 100.0))
print('WG start spread: %.2f us' % ((t[:, 0].max() - t0) / 100.0))
names = ['meta/table loads', 'tile zero + sync', 'particle loop (loads, constitutive, scan, LDS atomics) + sync', 'flush + sync']
for k in range(4):
    d = (t[:, k + 1] - t[:, k]) / 100.0
    print('%-70s mean %.2f us  p50 %.2f  p95 %.2f  max %.2f' % (names[k], d.mean(), np.median(d), np.percentile(d, 95), d.max()))
d = (t[:, 4] - t[:, 0]) / 100.0
print('whole item: mean %.2f p95 %.2f max %.2f us' % (d.mean(), np.percentile(d, 95), d.max()))
