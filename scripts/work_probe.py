"""The work list of the evolving benchmark block, window by window: items, pair units, small singles, quad units (fe_get_work_stats).
usage: python scripts/work_probe.py [n_windows] [opt=val ...]"""
import sys
sys.path.insert(0, '.')
import bench
from fluidlab_amd._capi import load_hip

n_win = int(sys.argv[1]) if len(sys.argv) > 1 else 35
eng, _ = bench.build_block(load_hip(), 0)
for o in sys.argv[2:]:
    k, v = o.split('='); eng.set_option(k, float(v))
print('win  items  multi-item-wgs leftovers singles small-singles pair-units  scatter-slots quad-units packed gather-slots active-blocks  by-size(1,2-4,5-8,9-16,17-32,33-64,65-128)')
for w in range(n_win):
    bench.window_step(eng, bench.CHUNK, backward=False)
    ws = eng.get_work_stats(0)
    nM, nS, nQ = ws['n_multi_item_workgroups'], ws['n_single_item_blocks'], ws['n_quad_items']
    print(f"{w:3d} {ws['n_items']:6d} {nM:8d} {ws['n_leftover_items']:8d} {nS:10d} {nQ:10d} {nM + (nS + 1) // 2:10d} {ws['n_scatter_units']:10d} {ws['n_quad_units']:8d} {int(ws['packed']):6d} {ws['n_gather_units']:10d} {ws['n_active_blocks']:10d}   {list(ws['items_by_size'].values())}")
eng.close()
