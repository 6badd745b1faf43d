"""In-kernel phase timeline of the six substep kernels on the benchmark scene (profiling build only).

  hipcc ... -DFE_TIMELINE -o fluidlab_amd/csrc/libfluidengine_tl_hip.so   (scripts/build_tl.sh)
  python scripts/timeline.py [n_grid n_particles [windows]]      windows: roll the block on by that many 24-substep windows first

Thread 0 of every workgroup stamps s_memrealtime (100 MHz) at the phase boundaries (TL(S, k) in fe_engine.hip).  Printed per
kernel: for each stamp k, the median / 90th percentile / max over workgroups of (stamp - earliest stamp 0 of the launch), in us.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidlab_amd import _capi  # noqa: E402
from fluidlab_amd.scenes import water_block, make_engine  # noqa: E402

PHASES = {
    'p2g': ['start', 'item', 'loaded+constitutive', 'barrier1', 'columns', 'barrier2', 'slab stored'],
    'grid_op': ['start', 'entry', 'slabs gathered', 'end'],
    'g2p': ['start', 'item', 'tile+barrier', 'particles', 'barrier'],
    'g2p_grad': ['start', 'item', 'tile+barrier', 'phase A', 'zero+barriers', 'columns', 'barrier', 'slab stored'],
    'p2g_grad': ['start', 'item', 'tile+barrier', 'particles', 'barrier'],
}
KNAMES = ['p2g', 'grid_op', 'g2p', 'p2g_recompute', 'grid_op_keep', 'g2p_grad', 'grid_op_grad', 'p2g_grad', 'sort', 'reorder_grad', 'sort_count', 'sort_scan', 'sort_active', 'sort_perm', 'g2p_p2g', 'pgg_g2pg']
# (FG kernels, option fuse_grid: stamp 3 = the workgroup's unit loop is over, 5 = wave 0 found an entry of its own complete (the last time), 4 = its entries worked on, 7 = their stores completed)
PHASES['g2p_p2g'] = ['start', 'item', 'loaded+constitutive', 'FG owner phase begins', 'FG entries done', 'FG entry found complete', 'slab stored', 'FG stores completed']
PHASES['pgg_g2pg'] = ['start', '1', '2', 'FG owner phase begins', 'FG entries done', 'FG entry found complete', '6', 'FG stores completed']


def main():
    n_grid = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    n_part = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
    elib = _capi.EngineLib(os.path.join(ROOT, 'fluidlab_amd', 'csrc', 'libfluidengine_tl_hip.so'))
    sc = water_block(n_grid=n_grid, n_particles=n_part, seed=0)
    L = 24
    eng = make_engine(elib, sc, max_substeps_local=L, device=0)
    for o in os.environ.get('TL_OPTS', '').split(','):
        if o:
            eng.set_option(o.split('=')[0], float(o.split('=')[1]))
    eng.loss_alloc(1)
    eng.loss_set_target(0, sc['x'])
    for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 0):         # let the block fall / splash first
        eng.step(0, 0, L, 0)
        eng.copy_frame(L, 0)
    for _ in range(2):
        eng.step(0, 0, L, 0)
        eng.reset_grad()
        eng.loss_step_grad(0, L, 0, 1.0, 1.0)
        eng.step_grad(0, 0, L, 0)
    eng.sync()
    st = eng.get_stats(L // 2)
    print('state:', {k: int(v) for k, v in st.items() if k != 'bytes_state'})
    buf = np.zeros((2048 * 9,), np.uint64)
    raw = {}
    for kid, name in enumerate(KNAMES):
        if name not in PHASES:
            continue
        rc = elib.lib.fe_timeline_read(eng.h, C.c_int(kid), buf.ctypes.data_as(C.c_void_p))
        assert rc == 0
        raw[name] = buf.copy()
        t = buf[:2048 * 8].reshape(2048, 8).astype(np.float64)
        ok = t[:, 0] > 0
        if not ok.any():
            continue
        t0 = t[ok, 0].min()
        print(f'== {name}: {int(ok.sum())} workgroups stamped; start spread {1e-2 * (t[ok, 0].max() - t0):.2f} us')
        for k, ph in enumerate(PHASES[name]):
            v = t[:, k]
            v = v[v > 0]
            if len(v) == 0:
                continue
            d = (v - t0) * 1e-2
            print(f'   {k} {ph:22s} n={len(v):5d}  median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us')
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, 'gpurun_out', f'timeline_raw_{n_grid}_{n_part}.npz'), **raw)
    eng.close()


if __name__ == '__main__':
    main()
