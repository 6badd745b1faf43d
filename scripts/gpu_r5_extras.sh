#!/bin/bash
# Round 5: the measurements beside the kernel A/B -- config 5 as SURVEY writes it, two config-3 replicas per GPU, the bench line with its extras.
# usage: scripts/gpu_r5_extras.sh <tag> [c5] [replicas] [bench] [ab configs...]
TAG=${1:-r5x}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
while [ $# -gt 0 ]; do
  case "$1" in
    c5)
      echo "== config 5 injected: parity test"; timeout 900 python -m pytest tests/test_hip_configs.py -k config5_injected -m gpu -q -s 2>&1 | grep -E "MEASURED|passed|failed|Error|assert " | cut -c1-900 | tee $OUT/pytest_c5.txt
      echo "== config 5 injected: dt 5e-5, 130 steps"; timeout 600 python scripts/run_c5.py 130 4 1000000 5e-5 2>&1 | grep -v amdgpu.ids | tee $OUT/run_c5_dt5e-5.txt | tail -4 | cut -c1-400
      echo "== config 5 injected: the reference's dt 2e-4"; timeout 300 python scripts/run_c5.py 60 4 1000000 2e-4 2>&1 | grep -v amdgpu.ids | tee $OUT/run_c5_dt2e-4.txt | tail -2 | cut -c1-300 ;;
    replicas)
      for B in 1 2; do
        echo "== config-3 replicas, B = $B per GPU"; timeout 900 python bench.py --gpus 1 --replicas --envs-per-gpu $B --steps 2 --warmup 1 2>$OUT/replica_config3_B$B.err | grep '^{"metric"' | tail -1 | tee $OUT/replica_config3_B$B.json | cut -c1-700; tail -3 $OUT/replica_config3_B$B.err | cut -c1-300
      done ;;
    bench)
      echo "== bench, default flags of the driver, with extras"; timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"' | tail -1 | tee $OUT/bench_driver_extras.json | cut -c1-300 ;;
    tests)
      echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu --maxfail=12 -q -s 2>&1 | grep -E "MEASURED|passed|failed|FAILED|Error|assert " | cut -c1-900 > $OUT/pytest_gpu.txt; tail -6 $OUT/pytest_gpu.txt ;;
    *)
      echo "== ab"; timeout 1200 python scripts/ab_phases.py --windows 35 --reps 2 "$@" 2>&1 | grep -v amdgpu.ids > $OUT/ab.txt; python scripts/ab_mean.py $OUT/ab.txt | tee $OUT/ab_mean.txt; break ;;
  esac
  shift
done
