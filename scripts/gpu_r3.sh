#!/bin/bash
# One GPU-box pass of round 3.  usage: scripts/gpu_r3.sh <tag> [tests|notests] [ab configs...]
TAG=${1:-r3}; shift
TESTS=${1:-tests}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m3 -E "Marketing Name|gfx9" > $OUT/device.txt
nproc >> $OUT/device.txt
if [ "$TESTS" = "tests" ]; then
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_random_scenes.py tests/test_kernel_golden.py tests/test_hip_fullsize.py tests/test_hip_env.py tests/test_smoke.py tests/test_mesh.py tests/test_hip_configs.py tests/test_bench.py -m gpu --maxfail=6 -q -s 2>&1 | grep -v "^$" | grep -v "^E  " | tail -70 | tee $OUT/pytest_gpu.txt
  echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
fi
if [ $# -gt 0 ]; then
  echo "== ab"; timeout 900 python scripts/ab_phases.py --reps 2 "$@" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
fi
echo "== bench (driver flags)"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -1 | tee $OUT/bench_driver.json
