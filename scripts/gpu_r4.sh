#!/bin/bash
# One GPU-box pass of round 4.  usage: scripts/gpu_r4.sh <tag> <tests|headline|notests> <windows> [ab configs...]
TAG=${1:-r4}; shift
TESTS=${1:-tests}; shift
WIN=${1:-35}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m3 -E "Marketing Name|gfx9" > $OUT/device.txt
nproc >> $OUT/device.txt
if [ "$TESTS" != "notests" ]; then
  echo "== headline parity"; timeout 1200 python -m pytest tests/test_hip_headline.py -m gpu -q -s 2>&1 | grep -E "MEASURED|passed|failed|Error|assert" | cut -c1-1500 | tee $OUT/pytest_headline.txt
fi
if [ "$TESTS" = "tests" ]; then
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu --deselect tests/test_hip_headline.py --maxfail=6 -q 2>&1 | grep -v "^$" | grep -v "^E  " | tail -30 | tee $OUT/pytest_gpu.txt
  echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
fi
echo "== work list"; timeout 300 python scripts/work_probe.py $WIN 2>&1 | grep -v amdgpu.ids | tee $OUT/work_probe.txt | tail -40
if [ $# -gt 0 ]; then
  echo "== ab"; timeout 1200 python scripts/ab_phases.py --windows $WIN --reps 2 "$@" 2>&1 | grep -v amdgpu.ids > $OUT/ab.txt; python scripts/ab_table.py $OUT/ab.txt | tee $OUT/ab_table.txt
fi
echo "== bench (driver flags)"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras 2>&1 | tail -1 | tee $OUT/bench_driver.json | cut -c1-1800
