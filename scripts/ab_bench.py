"""A/B: run bench.py's N=1 measurement with another build of the engine (same ABI), e.g. last round's kernels.
usage: python scripts/ab_bench.py fluidlab_amd/csrc/libfluidengine_r01_hip.so [bench.py flags]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidlab_amd import _capi  # noqa: E402

_capi.HIP_LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ['bench.py'] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
