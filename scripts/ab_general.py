"""The evolving ICECREAM block of bench.py's extra.general_128_200k (the SVD kernels at the metric's size) once, with whatever FE_* environment is set:
A/B of engine options on the GENERAL build, e.g. `for v in 1 0; do FE_FUSE_BWD=$v python scripts/ab_general.py; done`."""
import sys, json, os
sys.path.insert(0, '.')
import bench
from fluidlab_amd._capi import load_hip
from fluidlab_amd import scenes as S
lib = load_hip()
r = bench.extra_evolving(lib, 0, 'general', S.ICECREAM, 1e-4, 5, 25, '')
print(os.environ.get('FE_FUSE_BWD'), r.get('pairs_per_s'), {k: (v['avg_us'], v['launches']) for k, v in r.get('kernels', {}).items()}, r.get('error'))
