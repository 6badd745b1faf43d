"""Fold the per-kernel rocprofv3 --pmc summaries (scripts/gpu_pmc.sh, one pass per counter group) into
profiles/<tag>_pmc_traffic.json, the file bench.py reads `roofline.traffic` from.
usage: python scripts/make_pmc_traffic.py <tag> <fetch_summary> <write_summary> [<tcc_summary>]"""
import json
import sys

NAMES = {'void k_p2g<true, false>': 'p2g', 'void k_p2g<false, false>': 'p2g_recompute', 'void k_grid<false>': 'grid_op',
         'void k_grid<true>': 'grid_op_keep', 'k_g2p': 'g2p', 'k_g2p_grad': 'g2p_grad', 'k_grid_grad': 'grid_op_grad', 'void k_g2p<false>': 'g2p', 'void k_g2p_grad<false>': 'g2p_grad',
         'void k_grid<false, false>': 'grid_op', 'void k_grid<true, false>': 'grid_op_keep', 'void k_grid_grad<false>': 'grid_op_grad',
         'void k_grid<false, false, false>': 'grid_op', 'void k_grid<true, false, false>': 'grid_op_keep', 'void k_grid_grad<false, false>': 'grid_op_grad',
         'void k_p2g_grad<false, 4>': 'p2g_grad', 'void k_g2p_p2g<false>': 'g2p_p2g', 'void k_pgg_g2pg<4, false>': 'pgg_g2pg', 'void k_p2g<true, true>': 'p2g_general', 'void k_p2g_grad<true, 4>': 'p2g_grad_general'}


def parse(path):
    out = {}
    for line in open(path):
        parts = line.rstrip().split(' ')
        i = next(j for j, p in enumerate(parts) if '=' in p)
        name = ' '.join(parts[:i])
        out[name] = {k: float(v) for k, v in (p.split('=') for p in parts[i:])}
    return out


tag = sys.argv[1]
merged = {}
for path in sys.argv[2:]:
    for name, vals in parse(path).items():
        if name in NAMES:
            d = merged.setdefault(NAMES[name], {})
            for k, v in vals.items():
                d[{'FETCH_SIZE': 'FETCH_SIZE_KB', 'WRITE_SIZE': 'WRITE_SIZE_KB'}.get(k, k)] = v
for d in merged.values():
    if 'FETCH_SIZE_KB' in d and 'WRITE_SIZE_KB' in d:
        d['traffic_raw_bytes'] = int((d['FETCH_SIZE_KB'] + d['WRITE_SIZE_KB']) * 1024)
        # MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE tallies the 128-B requests of wide (16 B/lane) coalesced
        # reads at 64 B -> double it; WRITE_SIZE is taken as reported.  All frame/grid reads here are float4 loads.
        d['traffic_bytes'] = int((2 * d['FETCH_SIZE_KB'] + d['WRITE_SIZE_KB']) * 1024)
json.dump({'note': 'rocprofv3 --pmc per-launch averages over `bench.py --steps 2 --warmup 1` (water block 128^3 / 200k), one pass per '
                   'counter group (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum), --kernel-trace only.  traffic_bytes applies the '
                   'gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md; traffic_raw_bytes is the uncorrected reading.',
           'kernels': merged}, open(f'profiles/{tag}_pmc_traffic.json', 'w'), indent=1)
print(json.dumps(merged, indent=1))
