#!/bin/bash
# round 6: parity with the fused grid pass forced on, then an A/B over the phases of the benchmark block.  usage: WIN=35 scripts/gpu_r6a.sh <cfg> <cfg> ...
OUT=gpurun_out/r6a; mkdir -p $OUT; export TMPDIR=/tmp
PAR="tests/test_hip_parity.py tests/test_hip_random_scenes.py tests/test_kernel_golden.py"
echo "== parity, fuse_grid=2"; FE_FUSE_GRID=2 timeout 900 python -m pytest $PAR -m gpu --maxfail=10 -q 2>&1 | grep -E "passed|failed|FAILED|Error|assert " | cut -c1-400 | tee $OUT/pytest_fg2.txt | tail -15
echo "== ab"; timeout 1500 python scripts/ab_phases.py --windows ${WIN:-35} --reps ${REPS:-2} "$@" 2>&1 | grep -v amdgpu.ids > $OUT/ab.txt; python scripts/ab_table.py $OUT/ab.txt | tee $OUT/ab_table.txt
