"""pretty-print the json lines of scripts/ab_phases.py.  usage: ab_table.py <ab.txt>"""
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')]
ks = ['p2g', 'g2p_p2g', 'grid_op', 'g2p', 'g2p_grad', 'grid_op_grad', 'p2g_grad', 'pgg_g2pg', 'sort', 'reorder_grad', 'sort_count', 'sort_scan', 'sort_active', 'sort_perm']
for ph in ('falling', 'impact', 'esplash', 'timed', 'splash', 'layer', 'all'):
    print('==', ph)
    for r in rows:
        d = r.get(ph)
        if d:
            print(f"{r['config'][:44]:44s} {d['pairs_per_s'] or 0:8.0f} p/s {d['us_per_pair'] or 0:6.1f} us | " + ' '.join(f"{k[:8]}={d['us'][k]:5.1f}" for k in ks if k in d['us']))
