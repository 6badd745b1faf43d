"""Run a script of this repository with another build of the engine library (same ABI): python scripts/ab_run.py <lib.so> <script.py> [args]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidlab_amd import _capi  # noqa: E402

_capi.HIP_LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
