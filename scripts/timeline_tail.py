"""Which workgroups make the tail of a particle kernel?  Reads the raw stamps scripts/timeline.py saved (gpurun_out/timeline_raw_<n>_<N>.npz) and prints, for a kernel,
the distribution of every phase's LENGTH per workgroup, the end time by XCD (block % 8) and by dispatch order, and the phases of the slowest workgroups.
usage: python scripts/timeline_tail.py <npz> <kernel name> [...]"""
import sys
import numpy as np

z = np.load(sys.argv[1])
for name in sys.argv[2:]:
    t = z[name][:2048 * 8].reshape(2048, 8).astype(np.float64)
    hw = z[name][2048 * 8:]
    ok = t[:, 0] > 0
    idx = np.where(ok)[0]
    if not ok.any():
        print(f"== {name}: no stamps"); continue
    t0 = t[ok, 0].min()
    d = (t[ok] - t0) * 1e-2
    d[t[ok] == 0] = np.nan
    used = [k for k in range(8) if np.isfinite(d[:, k]).sum() > len(idx) // 2]
    end = np.nanmax(d, axis=1)
    print(f'== {name}: {len(idx)} workgroups; end median {np.median(end):.2f} p90 {np.percentile(end, 90):.2f} p99 {np.percentile(end, 99):.2f} max {end.max():.2f} us; stamps used {used}')
    for a, b in zip(used[:-1], used[1:]):
        seg = d[:, b] - d[:, a]
        seg = seg[np.isfinite(seg)]
        print(f'   phase {a}->{b}: median {np.median(seg):5.2f} p90 {np.percentile(seg, 90):5.2f} p99 {np.percentile(seg, 99):5.2f} max {seg.max():5.2f}')
    print('   end by XCD (block % 8):', ' '.join(f'{np.median(end[idx % 8 == x]):.1f}/{end[idx % 8 == x].max():.1f}' for x in range(8)))
    q = np.array_split(np.argsort(idx), 8)
    print('   end by dispatch order (eighths of the grid): ', ' '.join(f'{np.median(end[i]):.1f}/{end[i].max():.1f}' for i in q))
    slow = np.argsort(-end)[:12]
    for i in slow:
        print(f'   slow: block {idx[i]:4d} xcd {idx[i] % 8} ' + ' '.join(f'{d[i, k]:6.2f}' for k in used))
