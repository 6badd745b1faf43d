#!/bin/bash
# The round's profile evidence, on the GPU box: rocprofv3 kernel trace + PMC passes of the DRIVER'S command line, sliced into the
# phases of the evolving block (scripts/phase_profile.py).  usage: scripts/gpu_profile.sh <tag> [steps warmup]
TAG=${1:-r04}; STEPS=${2:-20}; WARM=${3:-5}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT/profiles
export TMPDIR=/tmp PROFILE_OUT=$OUT/profiles
CMD="python $PWD/bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-probe --no-extras --no-cpu-baseline"
echo "== kernel trace"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1)
KT=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python scripts/phase_profile.py stats $TAG "$KT" $STEPS $WARM | tee $OUT/phase_stats.txt
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/profiles/${TAG}_kernel_stats_whole_run.csv 2>/dev/null
PM=""
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  D=$OUT/pmc_$(echo $C | tr ' ' '_'); echo "== pmc $C"
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o p -- $CMD > $D.log 2>&1)
  F=$(find $D -name "*counter_collection.csv" | head -1); [ -n "$F" ] && PM="$PM $F"
done
python scripts/phase_profile.py pmc $TAG $STEPS $WARM $PM | tail -40 | tee $OUT/pmc_summary.txt
tail -2 $OUT/trace.log
# keep the merged-back payload small: the raw traces stay on the box
rm -rf $OUT/trace $OUT/pmc_*/
ls -la $OUT/profiles
