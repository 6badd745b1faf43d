"""How many blocks would a reach-limited active list hold?  (positions of the evolving block at a few times, counted on the host)"""
import sys, itertools
sys.path.insert(0, '.')
import numpy as np
import bench
from fluidlab_amd._capi import load_hip
from fluidlab_amd import scenes as S
eng, sc = bench.build_block(load_hip(), 0)
n = 128; nb = n // 4
for w in range(60):
    bench.window_step(eng, bench.CHUNK, backward=False)
    if w in (9, 17, 25, 33, 41, 59):
        x = S.get_state(eng, 0)['x']
        base = np.floor(x * n - 0.5).astype(np.int64)
        blk = base // 4; l = base - 4 * blk
        bid = (blk[:, 0] * nb + blk[:, 1]) * nb + blk[:, 2]
        occ = np.unique(bid)
        def nbrs(ids, deltas_of):
            out = set()
            bi, bj, bk = ids // (nb * nb), (ids // nb) % nb, ids % nb
            for t in range(len(ids)):
                for d in deltas_of(t):
                    i, j, k = bi[t] + d[0], bj[t] + d[1], bk[t] + d[2]
                    if 0 <= i < nb and 0 <= j < nb and 0 <= k < nb: out.add((i * nb + j) * nb + k)
            return out
        all27 = list(itertools.product((-1, 0, 1), repeat=3))
        full = nbrs(occ, lambda t: all27)
        # per-block axis flags: minus allowed if any particle has l == 0, plus if any has l >= 1
        order = np.argsort(bid, kind='stable'); sb = bid[order]; sl = l[order]
        first = np.searchsorted(sb, occ); last = np.searchsorted(sb, occ, side='right')
        minus = np.zeros((len(occ), 3), bool); plus = np.zeros((len(occ), 3), bool)
        for t in range(len(occ)):
            ll = sl[first[t]:last[t]]
            minus[t] = (ll == 0).any(0); plus[t] = (ll >= 1).any(0)
        def prod(t):
            ax = [([-1] if minus[t, a] else []) + [0] + ([1] if plus[t, a] else []) for a in range(3)]
            return itertools.product(*ax)
        axis = nbrs(occ, prod)
        # exact union of the particles' own 8 blocks
        ex = set()
        for a in itertools.product((0, 1), repeat=3):
            d = np.where(np.array(a)[None, :] == 1, np.where(l == 0, -1, 1), 0)
            q = blk + d
            ok = ((q >= 0) & (q < nb)).all(1)
            ex.update(((q[ok, 0] * nb + q[ok, 1]) * nb + q[ok, 2]).tolist())
        # blocks actually touched right now (no drift margin)
        tb = set()
        for a in itertools.product((0, 2), repeat=3):
            q = (base + np.array(a)) // 4
            tb.update(((q[:, 0] * nb + q[:, 1]) * nb + q[:, 2]).tolist())
        print(f'substep {(w + 1) * bench.CHUNK}: occupied {len(occ)}, 27-neighbourhood {len(full)}, axis-product reach {len(axis)}, exact reach union {len(ex)}, touched now {len(tb)}', flush=True)
