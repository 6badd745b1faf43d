"""Where the wall time of a rocprofv3 --kernel-trace run goes: kernel time by name and idle time, in equal slices of the span
between the first and the last substep kernel.  usage: trace_breakdown.py <..._kernel_trace.csv> [n_slices]"""
import collections
import csv
import re
import sys


def short(name):
    name = name.replace('void ', '')
    m = re.match(r'([A-Za-z_0-9:]+)(<[^>]*>)?', name)
    return (m.group(1) + (m.group(2) or ''))[:34] if m else name[:34]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    n_slices = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows))
    sub = [e for e in ev if e[2].startswith('k_p2g')]
    t0, t1 = sub[0][0], sub[-1][1]
    ev = [e for e in ev if e[0] >= t0 and e[1] <= t1]
    width = (t1 - t0) / n_slices
    print(f'span {1e-6 * (t1 - t0):.1f} ms, {len(ev)} launches, {n_slices} slices of {1e-6 * width:.1f} ms')
    for k in range(n_slices):
        lo, hi = t0 + k * width, t0 + (k + 1) * width
        busy = collections.Counter(); calls = collections.Counter()
        last_end, idle = lo, 0.0
        for s, e, nm in ev:
            if e <= lo or s >= hi:
                continue
            s2, e2 = max(s, lo), min(e, hi)
            busy[nm] += e2 - s2; calls[nm] += 1
            if s2 > last_end:
                idle += s2 - last_end
            last_end = max(last_end, e2)
        idle += max(0.0, hi - last_end)
        pairs = calls.get('k_p2g_grad<false, 4>', 0) or calls.get('k_g2p_grad', 0) or 1
        fwd = max(1, sum(v for n, v in calls.items() if n.startswith('k_p2g<')))
        print(f'-- slice {k}: {fwd} forward / {pairs} backward substeps, idle {100 * idle / width:.1f} % = {1e-3 * idle / max(fwd, pairs):.2f} us per substep')
        for nm, b in busy.most_common(14):
            print(f'   {nm:36s} {100 * b / width:5.1f} %  {calls[nm]:6d} launches  avg {1e-3 * b / calls[nm]:7.2f} us')


if __name__ == '__main__':
    main()
