"""Slice the rocprofv3 output of ONE bench.py run into the windows of the evolving block (window w = forward substeps
[100 w, 100 w + 100) and the backward sweep that follows them) and write per-phase summaries under profiles/:

  phase_profile.py stats <tag> <kernel_trace.csv> [steps warmup]
      -> profiles/<tag>_kernel_stats_<phase>.csv   (rocprofv3 --stats columns, one file per phase)
  phase_profile.py pmc <tag> <steps> <warmup> <counter_collection.csv> [<counter_collection.csv> ...]
      -> profiles/<tag>_pmc_traffic.json           (per phase and kernel: FETCH_SIZE / WRITE_SIZE per launch, gfx950-corrected traffic)

The run has to be `bench.py --steps S --warmup W --no-probe --no-extras --no-cpu-baseline` (one trajectory in the trace): the
timed region is windows [W, W + S).  Phases: the timed region itself, and fixed substep ranges of the block's history --
falling 500..1100, impact 1100..1800, splash 1800..4500, layer 4500.. -- as far as the run reaches them."""
import collections
import csv
import json
import math
import os
import sys

OUT = os.environ.get('PROFILE_OUT', 'profiles')          # (on the GPU box: a directory under gpurun_out/, copied into profiles/ afterwards)

CHUNK = 100
BENCH_NAME = [('k_p2g_grad', 'p2g_grad'), ('k_g2p_grad', 'g2p_grad'), ('k_grid_grad', 'grid_op_grad'), ('k_p2g<true', 'p2g'), ('k_p2g<false', 'p2g_recompute'),
              ('k_grid<false', 'grid_op'), ('k_grid<true', 'grid_op_keep'), ('k_g2p<', 'g2p'), ('k_g2p_sortkey', 'g2p'), ('k_g2p_p2g', 'g2p_p2g'), ('k_pgg_g2pg', 'pgg_g2pg')]
SORT_KERNELS = ('k_sort', 'k_scan', 'k_build', 'k_clear_slots', 'k_set_static', 'k_block')
FIXED = {'falling': (5, 11), 'impact': (11, 18), 'splash': (18, 45), 'layer': (45, 10**9)}


def _source_hash():
    """the engine build these numbers belong to (bench.py prints them only beside the same build on the same GPU)"""
    try:
        return open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'fluidlab_amd', 'csrc', 'libfluidengine_hip.so.srchash')).read().strip()
    except OSError:
        return None


def _device_name():
    try:
        import torch
        return torch.cuda.get_device_name(0)
    except Exception:
        return None


def clean(name):
    return name.replace('void ', '').strip()


def bench_name(name):
    n = clean(name)
    for pre, b in BENCH_NAME:
        if n.startswith(pre):
            return b
    return None


def windows(names):
    """window index of every launch of a run, in launch order"""
    out, n_fwd, forward = [], 0, True
    for nm in names:
        n = clean(nm)
        if n.startswith(('k_p2g<true', 'k_g2p_p2g')):          # a forward substep starts: its p2g, alone or behind the previous substep's g2p in one launch
            out.append(n_fwd // CHUNK); n_fwd += 1; forward = True
            continue
        if n.startswith(('k_g2p_grad', 'k_grid_grad', 'k_p2g_grad', 'k_pgg_g2pg', 'k_p2g<false', 'k_grid<true', 'k_perm_reorder', 'k_loss_bwd')):
            forward = False
        if forward and n.startswith(SORT_KERNELS):
            out.append(n_fwd // CHUNK)                       # the sort ahead of the substep that is about to start
        else:
            out.append(max(0, n_fwd - 1) // CHUNK)
    return out


def phases_of(steps, warmup, w_max):
    ph = {'timed_region': (warmup, warmup + steps)}
    for k, (a, b) in FIXED.items():
        if a <= w_max:
            ph[k] = (a, min(b, w_max + 1))
    return ph


def stats(tag, path, steps, warmup):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))
    wins = windows([r['Kernel_Name'] for r in rows])
    w_max = max(wins)
    for ph, (a, b) in phases_of(steps, warmup, w_max).items():
        dur = collections.defaultdict(list)
        for r, w in zip(rows, wins):
            if a <= w < b:
                dur[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        total = sum(sum(v) for v in dur.values()) or 1
        out = f'{OUT}/{tag}_kernel_stats_{ph}.csv'
        with open(out, 'w', newline='') as f:
            wr = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
            wr.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev'])
            for nm, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
                mean = sum(v) / len(v)
                sd = math.sqrt(sum((x - mean) ** 2 for x in v) / len(v))
                wr.writerow([nm, len(v), sum(v), round(mean, 3), round(100 * sum(v) / total, 2), min(v), max(v), round(sd, 3)])
        pairs = sum(len(v) for nm, v in dur.items() if clean(nm).startswith(('k_p2g_grad', 'k_pgg_g2pg')))      # a backward substep has one p2g_grad, alone or at the head of a k_pgg_g2pg launch
        print(f'{out}: windows [{a},{b}) = substeps {a * CHUNK}..{b * CHUNK}, {pairs} backward substeps, kernel time {1e-3 * total / max(1, pairs):.1f} us per pair')
        for nm, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:8]:
            print(f'    {clean(nm)[:44]:46s} {len(v):6d} x {1e-3 * sum(v) / len(v):7.2f} us')


def pmc(tag, steps, warmup, paths):
    per = {}                                                 # phase -> kernel -> counter -> [sum, n]
    w_max = 0
    for path in paths:
        rows = list(csv.DictReader(open(path)))
        by_disp = collections.OrderedDict()
        for r in sorted(rows, key=lambda r: int(r['Dispatch_Id'])):
            by_disp.setdefault(int(r['Dispatch_Id']), (r['Kernel_Name'], {}))[1][r['Counter_Name']] = float(r['Counter_Value'])
        disp = list(by_disp.values())
        wins = windows([d[0] for d in disp])
        w_max = max(w_max, max(wins))
        for ph, (a, b) in phases_of(steps, warmup, max(wins)).items():
            for (nm, ctr), w in zip(disp, wins):
                bn = bench_name(nm)
                if bn is None or not (a <= w < b):
                    continue
                for c, v in ctr.items():
                    acc = per.setdefault(ph, {}).setdefault(bn, {}).setdefault(c, [0.0, 0])
                    acc[0] += v; acc[1] += 1
    out = {'note': 'rocprofv3 --pmc per-launch averages of `bench.py --steps %d --warmup %d --no-probe --no-extras --no-cpu-baseline`, one pass per counter '
                   'group, --kernel-trace only, launches assigned to windows of 100 substeps (scripts/phase_profile.py).  traffic_bytes = 2 x FETCH_SIZE + '
                   'WRITE_SIZE (KB -> bytes): the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md for 16 B/lane reads; traffic_raw_bytes is the plain sum.' % (steps, warmup),
           'steps': steps, 'warmup': warmup, 'source_hash': _source_hash(), 'device': _device_name()}
    for ph, kern in per.items():
        a, b = phases_of(steps, warmup, w_max)[ph]
        d = {'substeps': [a * CHUNK, b * CHUNK], 'kernels': {}}
        for bn, ctr in kern.items():
            e = {('%s_KB' % c if c in ('FETCH_SIZE', 'WRITE_SIZE') else c): round(s / n, 1) for c, (s, n) in ctr.items()}
            e['launches'] = max(n for _, n in ctr.values())
            if 'FETCH_SIZE_KB' in e and 'WRITE_SIZE_KB' in e:
                e['traffic_raw_bytes'] = int((e['FETCH_SIZE_KB'] + e['WRITE_SIZE_KB']) * 1024)
                e['traffic_bytes'] = int((2 * e['FETCH_SIZE_KB'] + e['WRITE_SIZE_KB']) * 1024)
            d['kernels'][bn] = e
        out[ph] = d
    json.dump(out, open(f'{OUT}/{tag}_pmc_traffic.json', 'w'), indent=1)
    print(json.dumps({ph: {k: v.get('traffic_bytes') for k, v in out[ph]['kernels'].items()} for ph in per}, indent=1))


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 20, int(sys.argv[5]) if len(sys.argv) > 5 else 5)
    else:
        pmc(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5:])
