#!/bin/bash
# A short GPU-box pass for one experiment: selected parity tests, then an in-process A/B of option sets on the evolving block.
# usage: scripts/gpu_try.sh <tag> "<pytest -k expression>" <windows> [ab configs...]
TAG=${1:-try}; shift
KEXPR=${1:-fused}; shift
WIN=${1:-35}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -k '$KEXPR'"; timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_random_scenes.py -m gpu -k "$KEXPR" --maxfail=20 -q -s 2>&1 | grep -E "MEASURED|passed|failed|FAILED|Error|assert |^E " | cut -c1-700 > $OUT/pytest_gpu.txt; tail -40 $OUT/pytest_gpu.txt
if [ $# -gt 0 ]; then
  echo "== ab"; timeout 1200 python scripts/ab_phases.py --windows $WIN --reps 2 "$@" 2>&1 | grep -v amdgpu.ids > $OUT/ab.txt; python scripts/ab_mean.py $OUT/ab.txt | tee $OUT/ab_mean.txt
fi
