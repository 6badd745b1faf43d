"""The evolving block of bench.py over time: every `every` windows, the work list and the per-kernel times of one profiled window.
usage: evolve_probe.py [windows] [every] [opt=value ...]"""
import json, sys, time
sys.path.insert(0, '.')
import bench
from fluidlab_amd._capi import load_hip
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 100
every = int(sys.argv[2]) if len(sys.argv) > 2 else 10
eng, sc = bench.build_block(load_hip(), 0)
for o in sys.argv[3:]:
    k, v = o.split('='); eng.set_option(k, float(v))
for w in range(nw):
    if w % every == 0:
        eng.sync(); t0 = time.perf_counter()
        bench.window_step(eng, bench.CHUNK)
        eng.sync(); dt = time.perf_counter() - t0
        eng.profile_enable(True); bench.window_step(eng, bench.CHUNK); prof = eng.profile_read(); eng.profile_enable(False)
        st = eng.get_stats(0); ws = eng.get_work_stats(0)
        print(json.dumps({'substep': (w + 2) * bench.CHUNK, 'us_per_pair': round(1e6 * dt / bench.CHUNK, 1), 'nc': st['n_cells_touched'], 'slow': st['n_slow_path'],
                          **{k: v for k, v in ws.items()}, 'us': {k: round(1e3 * v[0] / v[1], 1) for k, v in prof.items() if v[1]}}))
    else:
        bench.window_step(eng, bench.CHUNK)
