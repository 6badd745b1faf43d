"""A/B of engine options on the evolving block of bench.py, phase by phase, in ONE process on one box.
usage: ab_phases.py [--windows N] [--reps R] "opt=val,opt=val" "opt=val" ...      ('' = defaults)
Every configuration replays the same N windows (default 90: falling 500-1100, splash 2600-3400, layer 8000-9000); per phase the
wall-clock rate of the plain windows and the HIP-event time per kernel of the profiled ones (bench.probe_pass / fold_windows)."""
import json, sys, time
sys.path.insert(0, '.')
import bench
from fluidlab_amd._capi import load_hip, EngineLib

args = sys.argv[1:]
n_win, reps = 90, 1
while args and args[0].startswith('--'):
    if args[0] == '--windows': n_win = int(args[1])
    if args[0] == '--reps': reps = int(args[1])
    args = args[2:]
configs = args or ['']
default_lib = load_hip()
libs = {}
orig = bench.build_block


def run(cfg):
    """cfg: comma-separated engine options; `lib=<path>` picks another build of the engine (A/B against an earlier round's kernels;
    options that build does not know are skipped)"""
    opts = [o.split('=') for o in cfg.split(',') if o]
    lib_path = next((v for k, v in opts if k == 'lib'), None)
    opts = [(k, v) for k, v in opts if k != 'lib']
    if lib_path and lib_path not in libs:
        libs[lib_path] = EngineLib(lib_path)
    elib = libs[lib_path] if lib_path else default_lib

    def build(*a, **k):
        eng, sc = orig(*a, **k)
        for name, v in opts:
            try:
                eng.set_option(name, float(v))
            except Exception as e:                  # an older build without that option
                print('   (option skipped:', name, e, ')', file=sys.stderr)
        return eng, sc
    bench.build_block = build
    try:
        profiled = {w for w in range(n_win) if w % 2 == 0}
        t0 = time.perf_counter()
        rec = bench.probe_pass(elib, n_win, range(0), profiled)
        wall = time.perf_counter() - t0
    finally:
        bench.build_block = orig
    out = {'config': cfg or '(defaults)', 'wall_s': round(wall, 2)}
    for name, (a, b) in list(bench.PHASES.items()) + [('esplash', (18, 25)), ('timed', (5, 25)), ('all', (5, n_win))]:
        if b <= n_win or name == 'all':
            f = bench.fold_windows(rec, a, min(b, n_win))
            out[name] = {'pairs_per_s': f.get('pairs_per_s'), 'us_per_pair': round(1e6 / f['pairs_per_s'], 1) if f.get('pairs_per_s') else None, 'nc': f['nc_mean'],
                         'us': {k: v['avg_us'] for k, v in f['kernels'].items()}}
    return out


for r in range(reps):
    for cfg in configs:
        print(json.dumps(run(cfg)), flush=True)
