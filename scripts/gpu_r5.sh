#!/bin/bash
# One GPU-box pass of round 5.  usage: scripts/gpu_r5.sh <tag> <tests|parity|notests> <windows> [ab configs...]
# tests: the whole -m gpu suite (MEASURED lines kept); on a failure the kernel-parity files are run again per A/B build / option so
# that one call says which of the round's changes broke them.
TAG=${1:-r5}; shift
TESTS=${1:-tests}; shift
WIN=${1:-35}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m3 -E "Marketing Name|gfx9" > $OUT/device.txt
nproc >> $OUT/device.txt
PAR="tests/test_hip_parity.py tests/test_hip_random_scenes.py tests/test_kernel_golden.py"
if [ "$TESTS" = "tests" ]; then
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu --maxfail=12 -q -s 2>&1 | grep -E "MEASURED|passed|failed|FAILED|Error|assert " | cut -c1-900 > $OUT/pytest_gpu.txt; tail -25 $OUT/pytest_gpu.txt
  echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
elif [ "$TESTS" = "parity" ]; then
  echo "== parity"; timeout 900 python -m pytest $PAR -m gpu --maxfail=12 -q -s 2>&1 | grep -E "MEASURED|passed|failed|FAILED|Error|assert " | cut -c1-900 > $OUT/pytest_gpu.txt; tail -25 $OUT/pytest_gpu.txt
fi
if [ "$TESTS" != "notests" ] && grep -q "failed" $OUT/pytest_gpu.txt; then
  echo "== bisect: lane_split off"; FE_LANE_SPLIT=0 timeout 600 python -m pytest $PAR -m gpu --maxfail=20 -q 2>&1 | grep -E "passed|failed|FAILED" | cut -c1-300 | tee $OUT/bisect_nosplit.txt | tail -12
  for V in nolean; do
    [ -f scripts/_bin/libfe_$V.so ] || continue
    echo "== bisect: build $V"; FE_TEST_HIP_LIB=scripts/_bin/libfe_$V.so timeout 600 python -m pytest $PAR -m gpu --maxfail=20 -q --deselect tests/test_hip_parity.py::test_native_library_is_the_one_loaded 2>&1 | grep -E "passed|failed|FAILED" | cut -c1-300 | tee $OUT/bisect_$V.txt | tail -12
    echo "== bisect: build $V, lane_split off"; FE_LANE_SPLIT=0 FE_TEST_HIP_LIB=scripts/_bin/libfe_$V.so timeout 600 python -m pytest $PAR -m gpu --maxfail=20 -q --deselect tests/test_hip_parity.py::test_native_library_is_the_one_loaded 2>&1 | grep -E "passed|failed|FAILED" | cut -c1-300 | tee $OUT/bisect_${V}_nosplit.txt | tail -12
  done
fi
if [ $# -gt 0 ]; then
  echo "== ab"; timeout 1200 python scripts/ab_phases.py --windows $WIN --reps 2 "$@" 2>&1 | grep -v amdgpu.ids > $OUT/ab.txt; python scripts/ab_table.py $OUT/ab.txt | tee $OUT/ab_table.txt
fi
echo "== bench (driver flags)"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras 2>&1 | tail -1 | tee $OUT/bench_driver.json | cut -c1-1500
if [ -n "$C5_STEPS" ]; then
  echo "== config 5 as SURVEY writes it (256^3, 1M pool, BallInjector)"; timeout 900 python scripts/run_c5.py $C5_STEPS 2>&1 | grep -v amdgpu.ids | tee $OUT/run_c5.txt | tail -20
fi
