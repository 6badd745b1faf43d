"""A/B of engine builds on bench.py's 256^3 / 1M-particle block (extra.config5_water_256_1M), in one process on one box.
usage: ab_1m.py [--reps R] [water|icecream] "" "lib=scripts/_bin/libfe_X.so" "opt=val,..." ...      ('' = the shipped build, defaults)"""
import sys
sys.path.insert(0, '.')
import bench
from fluidlab_amd import scenes as S
from fluidlab_amd._capi import load_hip, EngineLib

args = sys.argv[1:]
reps = 2
if args and args[0] == '--reps':
    reps = int(args[1]); args = args[2:]
mat = S.WATER
if args and args[0] in ('water', 'icecream'):
    mat = S.ICECREAM if args[0] == 'icecream' else S.WATER; args = args[1:]
default = load_hip()
orig = S.make_engine
for r in range(reps):
    for cfg in (args or ['']):
        opts = [o.split('=') for o in cfg.split(',') if o]
        lib = next((v for k, v in opts if k == 'lib'), None)
        elib = EngineLib(lib) if lib else default

        def make(*a, **k):
            eng = orig(*a, **k)
            for name, v in opts:
                if name != 'lib':
                    eng.set_option(name, float(v))
            return eng
        S.make_engine = make
        try:
            o = bench.extra_block(elib, 0, 'ab', 256, 1_000_000, mat, 10, 6)
        finally:
            S.make_engine = orig
        print(f"{cfg or '(shipped)':44s} {o['pairs_per_s']:8.1f} p/s  frac {o['pair_roofline']['frac']:.4f} | " + ' '.join(f'{k}={v}' for k, v in o['kernels_us'].items()), flush=True)
