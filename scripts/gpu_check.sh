#!/bin/bash
# One GPU-box pass: parity tests, smoke, bench, rocprof kernel trace.  Outputs under gpurun_out/.
# usage: scripts/gpu_check.sh [tag]
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m3 -E "Marketing Name|gfx9" > $OUT/device.txt
nproc >> $OUT/device.txt; grep -m1 "model name" /proc/cpuinfo >> $OUT/device.txt
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 | tee $OUT/pytest_gpu.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.txt
echo "== bench"; timeout 900 python bench.py 2>&1 | tail -5 | tee $OUT/bench.json
echo "== rocprof"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o trace -- python $OLDPWD/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $OLDPWD/$OUT/rocprof_bench.log 2>&1)
find $OUT/prof -name "*kernel_stats*" | head -3
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" | tee $OUT/kernel_stats_head.csv
# keep the merged-back payload small
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
