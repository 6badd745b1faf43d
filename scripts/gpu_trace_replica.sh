#!/bin/bash
# rocprofv3 kernel stats of one LatteArt replica pass (as shipped, 64^3): which kernels fluidlab's step / step_grad flow launches, how often.  usage: scripts/gpu_trace_replica.sh <tag>
TAG=${1:-trr}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python $PWD/bench.py --gpus 1 --replicas --c4-scene as_shipped --steps 1 --warmup 0"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1)
F=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cp $F $OUT/kernel_stats_replica_as_shipped.csv; rm -rf $OUT/trace
python - $OUT/kernel_stats_replica_as_shipped.csv <<'P'
import csv, sys
for r in csv.reader(open(sys.argv[1])):
    if r[0] != 'Name': print(r[0][:60].ljust(60), r[1].rjust(7), 'x', f'{float(r[3]) / 1000:8.2f} us', r[4], '%')
P
