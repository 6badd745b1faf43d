#!/bin/bash
# run selected tests with full assertion output.  usage: scripts/gpu_t.sh <tag> <pytest args...>
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1500 python -m pytest "$@" -m gpu -q -s 2>&1 | grep -v "^$" | cut -c1-600 | grep -v "^    \|^hiplib\|^oracle" | tail -150 | tee $OUT/pytest.txt
