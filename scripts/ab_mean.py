"""mean over the repetitions of scripts/ab_phases.py's json lines, one row per configuration and phase.  usage: ab_mean.py <ab.txt>"""
import collections, json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')]
for ph in ('falling', 'impact', 'esplash', 'timed', 'splash', 'layer', 'all'):
    agg = collections.OrderedDict()
    for r in rows:
        d = r.get(ph)
        if d:
            agg.setdefault(r['config'], []).append(d)
    if not agg:
        continue
    print('==', ph)
    for c, ds in agg.items():
        us = sum(d['us_per_pair'] for d in ds) / len(ds)
        ks = {k: sum(d['us'].get(k, 0.0) for d in ds) / len(ds) for k in ds[0]['us']}
        print(f"{c[:40]:40s} {1e6 / us:7.0f} p/s {us:7.1f} us | " + ' '.join(f"{k[:8]}={v:5.1f}" for k, v in ks.items() if k in ('p2g', 'g2p_p2g', 'grid_op', 'g2p', 'g2p_grad', 'grid_op_grad', 'p2g_grad', 'pgg_g2pg', 'sort')))
