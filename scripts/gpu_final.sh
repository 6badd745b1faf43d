#!/bin/bash
# End-of-round evidence besides the profiles: the bench lines the driver will reproduce, the replica path.  usage: scripts/gpu_final.sh <tag>
TAG=${1:-r04f}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m3 -E "Marketing Name|gfx9" > $OUT/device.txt; nproc >> $OUT/device.txt; grep -m1 "model name" /proc/cpuinfo >> $OUT/device.txt
echo "== bench (driver flags)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | grep '^{' | tail -1 > $OUT/bench_driver.json; cut -c1-300 $OUT/bench_driver.json
echo "== bench (defaults)"; timeout 900 python bench.py --no-extras 2>&1 | grep '^{' | tail -1 > $OUT/bench_default.json; cut -c1-300 $OUT/bench_default.json
echo "== replica, config 3, one rank over RCCL"
for TRY in 1 2; do      # (the first process on a fresh box to bring up RCCL has come back without a line once: keep its log, try again)
  timeout 1200 python bench.py --gpus 1 --replicas --steps 2 --warmup 1 > $OUT/replica_config3.log 2>&1; grep '^{' $OUT/replica_config3.log | tail -1 > $OUT/replica_config3.json
  [ -s $OUT/replica_config3.json ] && break; tail -5 $OUT/replica_config3.log
done
cut -c1-400 $OUT/replica_config3.json
echo "== replicas, as shipped, B = 1 / 2 / 3 per GPU"
for B in 1 2 3; do timeout 1200 python bench.py --gpus 1 --replicas --c4-scene as_shipped --envs-per-gpu $B --steps 3 --warmup 1 2>&1 | grep '^{' | tail -1 > $OUT/replica_as_shipped_B$B.json; cut -c1-200 $OUT/replica_as_shipped_B$B.json; done
