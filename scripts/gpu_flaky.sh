#!/bin/bash
# how reproducible is a test?  usage: scripts/gpu_flaky.sh <tag> <n> <test id> [alt lib]
TAG=$1; N=$2; T=$3; ALT=$4
OUT=gpurun_out/$TAG; mkdir -p $OUT
for i in $(seq 1 $N); do
  python -m pytest "$T" -m gpu -q -s 2>&1 | grep -E "MEASURED|passed|failed" | cut -c1-260 | sed "s/^/new $i: /" | tee -a $OUT/flaky.txt
  if [ -n "$ALT" ]; then FE_TEST_HIP_LIB=$ALT python -m pytest "$T" -m gpu -q -s 2>&1 | grep -E "MEASURED|passed|failed" | cut -c1-260 | sed "s/^/alt $i: /" | tee -a $OUT/flaky.txt; fi
done
