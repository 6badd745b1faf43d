#!/bin/bash
# parity tests of the kernels + an A/B of engine builds / options over the phases of the benchmark block.  usage: WIN=46 scripts/gpu_ab.sh <config> [<config> ...]
OUT=gpurun_out/ab; mkdir -p $OUT; export TMPDIR=/tmp
WIN=${WIN:-30}
echo "== tests"; timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_random_scenes.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest.txt
echo "== ab"; timeout 1200 python scripts/ab_phases.py --windows $WIN --reps 2 "$@" 2>&1 | grep -v amdgpu.ids > $OUT/ab.txt; python scripts/ab_table.py $OUT/ab.txt | tee $OUT/ab_table.txt
