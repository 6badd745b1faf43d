"""How much of the scatter kernels' LDS-atomic work is the stale order?  Host-side count of the runs of equal stencil base the DPP
scan sees (runs are cut at the 16-lane rows; one ds_add per run, node and value), for the benchmark block at a given window:
  as sorted `age` substeps ago (what the kernels see)  |  lanes re-sorted inside every wave  |  inside every item  |  a fresh sort
for two intra-block cell orders of the sort key (z fastest = the grid layout, y fastest).
usage: python scripts/run_stats.py <window> [age ...]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import bench
from fluidlab_amd import scenes as S
from fluidlab_amd._capi import load_hip

win = int(sys.argv[1]) if len(sys.argv) > 1 else 13
ages = [int(a) for a in sys.argv[2:]] or [1, 3, 5, 9]
eng, _ = bench.build_block(load_hip(), 0)
for w in range(win):
    bench.window_step(eng, bench.CHUNK, backward=False)
eng.step(0, 0, max(ages) + 1, 0)
n = bench.N_GRID


def base(x):
    return np.floor(x.astype(np.float32) * np.float32(n) - np.float32(0.5)).astype(np.int64)


def keys(b, order):
    blk = ((b[:, 0] >> 2) * (n // 4) + (b[:, 1] >> 2)) * (n // 4) + (b[:, 2] >> 2)
    i, j, k = b[:, 0] & 3, b[:, 1] & 3, b[:, 2] & 3
    cell = (i << 4) | (j << 2) | k if order == 'z' else (i << 4) | (k << 2) | j
    return blk, blk * 64 + cell


def runs(key, valid):
    """number of runs of equal adjacent keys, cut at 16-lane rows; key: [n_waves, 64]"""
    k = np.where(valid, key, -1 - np.arange(key.size).reshape(key.shape))
    head = np.ones(k.shape, bool)
    head[:, 1:] = k[:, 1:] != k[:, :-1]
    head[:, ::16] = True
    return int((head & valid).sum())


x0 = S.get_state(eng, 0)['x']
for order in ('z', 'y'):
    blk0, key0 = keys(base(x0), order)
    perm = np.argsort(key0, kind='stable')                 # the sort at frame 0 (within a cell: arbitrary, here by particle id)
    # items: <= 128 consecutive slots of one block; every item starts a wave
    b_sorted = blk0[perm]
    starts = np.flatnonzero(np.r_[True, b_sorted[1:] != b_sorted[:-1]])
    lens = np.diff(np.r_[starts, len(perm)])
    slots = []
    for s, l in zip(starts, lens):
        for o in range(0, l, 128):
            m = min(128, l - o)
            pad = (-m) % 64
            slots.append(np.r_[perm[s + o:s + o + m], np.full(pad, -1)])
    lanes = np.concatenate(slots).reshape(-1, 64)          # particle id per lane, -1 = idle lane
    valid = lanes >= 0
    print(f'== window {win}, intra-block cell order {order}-fastest: {len(slots)} items, {lanes.shape[0]} waves, {valid.sum()} particles')
    for age in [0] + ages:
        xa = S.get_state(eng, age)['x']
        blk_a, key_a = keys(base(xa), order)
        ka = np.where(valid, key_a[np.maximum(lanes, 0)], 1 << 60)
        as_is = runs(ka, valid)
        in_wave = runs(np.sort(ka, axis=1), valid)
        # inside every item: sort the item's 128 lanes (two waves) together
        item_rows = []
        r = 0
        for sl in slots:
            w = len(sl) // 64
            kk = np.sort(ka[r:r + w].reshape(-1))
            item_rows.append(kk.reshape(w, 64)); r += w
        in_item = runs(np.concatenate(item_rows), valid)    # (valid lanes sort to the front of each item: the mask still fits)
        fresh_key = np.sort(key_a)
        fresh = int((np.r_[True, fresh_key[1:] != fresh_key[:-1]] | (np.arange(len(fresh_key)) % 16 == 0)).sum())
        left_block = float((blk_a != blk0).mean())
        print(f'   age {age:2d}: runs as-is {as_is:7d}   in-wave sort {in_wave:7d}   in-item sort {in_item:7d}   fresh sort ~{fresh:7d}   (particles that left their block: {100 * left_block:.1f} %)')
eng.close()
