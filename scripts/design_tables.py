"""The per-kernel tables of DESIGN.md sections 5 / 6 from a round's committed evidence.  usage: design_tables.py r06 [r05]   (the second tag: figures in brackets)
Reads profiles/<tag>_kernel_stats_{timed_region,falling,impact,splash}.csv, <tag>_pmc_traffic.json, <tag>_pmc_issue_counters.txt, <tag>_bench_driver_flags.json."""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

SHORT = {'k_pgg_g2pg<4, false>': 'pgg_g2pg', 'k_g2p_p2g<false>': 'g2p_p2g', 'k_grid_grad<false, false>': 'grid_op_grad', 'k_grid<false, false, false>': 'grid_op',
         'k_g2p_grad2<4>': 'g2p_grad', 'k_p2g<true, false>': 'p2g', 'k_p2g_grad<false, 4>': 'p2g_grad', 'k_g2p<false>': 'g2p', 'k_g2p_sortkey<false>': 'g2p'}
SORT = ('k_sort_count', 'k_sort_blk_partial', 'k_sort_blk_final', 'k_sort_blk_scan', 'k_sort_apply')


def stats(tag, phase):
    p = os.path.join(ROOT, 'profiles', f'{tag}_kernel_stats_{phase}.csv')
    out = {}
    if not os.path.exists(p):
        return out
    for r in csv.DictReader(open(p)):
        n = r['Name'].replace('void ', '').split('(')[0]
        out[n] = (int(r['Calls']), float(r['AverageNs']) / 1e3)
    return out


def main():
    tag = sys.argv[1]
    prev = sys.argv[2] if len(sys.argv) > 2 else None
    b = json.load(open(os.path.join(ROOT, 'profiles', f'{tag}_bench_driver_flags.json')))
    N, Nc = b['config']['n_used'], b['config']['nc_mean_timed']
    pm = json.load(open(os.path.join(ROOT, 'profiles', f'{tag}_pmc_traffic.json')))['timed_region']['kernels']
    issue = {}
    ic = os.path.join(ROOT, 'profiles', f'{tag}_pmc_issue_counters.txt')
    if os.path.exists(ic):
        for l in open(ic):
            m = re.match(r'(?:void )?(\S+?)[(<].*VALU issue time ([\d.]+) us', l)
            if m:
                issue.setdefault(l.split('(')[0].replace('void ', '').strip()[:20], float(m.group(2)))
    st, sp = stats(tag, 'timed_region'), stats(prev, 'timed_region') if prev else {}
    pairs = st['k_grid_grad<false, false>'][0]
    print(f'N = {N}, Nc = {Nc}, backward substeps {pairs}\n')
    print('| kernel (timed region) | launches per pair | rocprofv3 avg us | algorithmic MB | GB/s | frac of 8 TB/s | counted traffic MB (/ algorithmic) | VALU issue us (share) |')
    print('|---|---|---|---|---|---|---|---|')
    tot_t = 0.0
    for k, (calls, us) in sorted(st.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        tot_t += calls * us
        if k not in SHORT:
            continue
        s = SHORT[k]
        bp, bc = bench.KERNEL_BYTES[s]
        alg = bp * N + bc * Nc + (bench.STATE_REREAD * N if s in ('pgg_g2pg', 'p2g_grad') else 0)
        tr = pm.get(s)
        trb = tr.get('traffic_bytes') if tr else None
        iss = next((v for kk, v in issue.items() if k.startswith(kk[:18]) or kk.startswith(k[:18])), None)
        pv = f' ({sp[k][1]:.2f})' if k in sp else ''
        print(f'| `{k}` | {calls / pairs:.2f} | {us:.2f}{pv} | {alg / 1e6:.1f} | {alg / us / 1e3:,.0f} | {alg / us / 1e3 / 8000:.3f} | '
              + (f'{trb / 1e6:.1f} ({trb / alg:.2f}x)' if trb else '--') + ' | ' + (f'{iss:.1f} ({100 * iss / us:.0f} %)' if iss else '--') + ' |')
    srt = sum(c * u for k, (c, u) in st.items() if k in SORT)
    ns = st.get('k_sort_apply', (1, 0))[0]
    print(f'| sort ({", ".join(k for k in SORT if k in st)}) | {ns / pairs:.2f} | {srt / ns:.1f} per event' + (f' ({sum(c * u for k, (c, u) in sp.items() if k in SORT) / sp["k_sort_apply"][0]:.1f})' if sp else '') + ' | 0 | | | | |')
    alg_pair = b['pair_roofline']['alg_bytes_per_pair']
    print(f'| **pair** | | **{tot_t / pairs:.1f}** (trace) / {1e6 / b["value"]:.1f} (driver clock) | {alg_pair / 1e6:.1f} | {alg_pair * b["value"] / 1e9:,.0f} | **{b["pair_roofline"]["frac"]:.4f}** | | |')
    print()
    print('| | `k_g2p_p2g` | `k_grid` | `k_pgg_g2pg` | `k_grid_grad` | separate: p2g / g2p (with keys) / g2p_grad2 / p2g_grad | sort per event | kernel time per pair |')
    print('|---|---|---|---|---|---|---|---|')
    for ph in ('timed_region', 'falling', 'impact', 'splash'):
        s, q = stats(tag, ph), stats(prev, ph) if prev else {}
        if not s:
            continue
        g = lambda k, d=s: f'{d[k][1]:.2f}' if k in d else '--'
        npair = s['k_grid_grad<false, false>'][0]
        tot = sum(c * u for c, u in s.values()) / npair
        totq = sum(c * u for c, u in q.values()) / q['k_grid_grad<false, false>'][0] if q else None
        so = sum(c * u for k, (c, u) in s.items() if k in SORT) / max(1, s.get('k_sort_apply', (1, 0))[0])
        print(f'| {ph.replace("_", " ")} | {g("k_g2p_p2g<false>")} | {g("k_grid<false, false, false>")} | {g("k_pgg_g2pg<4, false>")} | {g("k_grid_grad<false, false>")} | '
              f'{g("k_p2g<true, false>")} / {g("k_g2p_sortkey<false>")} / {g("k_g2p_grad2<4>")} / {g("k_p2g_grad<false, 4>")} | {so:.1f} | **{tot:.1f}**' + (f' ({totq:.1f})' if totq else '') + ' |')


if __name__ == '__main__':
    main()
