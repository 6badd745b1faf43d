#!/bin/bash
# short rocprofv3 kernel trace of the driver's command (stats only) -> gpurun_out/<tag>/phase_stats.txt.  usage: scripts/gpu_trace_short.sh <tag> [steps warmup]
TAG=${1:-tr}; STEPS=${2:-20}; WARM=${3:-5}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT/profiles
export TMPDIR=/tmp PROFILE_OUT=$OUT/profiles
CMD="python $PWD/bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-probe --no-extras --no-cpu-baseline"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1)
KT=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python scripts/phase_profile.py stats $TAG "$KT" $STEPS $WARM > $OUT/phase_stats.txt
grep -A 14 "timed_region" $OUT/phase_stats.txt | cut -c1-110
tail -1 $OUT/trace.log | cut -c1-300
rm -rf $OUT/trace
