#!/bin/bash
# usage: scripts/gpu_sweep.sh tag "optA=1 optB=2" "optA=3" ...   (each arg = one bench run with those engine options)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for cfg in "$@"; do
  args=""; for o in $cfg; do args="$args --opt $o"; done
  echo "== $cfg"
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-replica-probe $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
print('pairs/s', d['value'], 'fwd/s', d['forward_only_substeps_per_s'], ' '.join(f\"{n}={v['avg_us']:.1f}\" for n,v in k.items()))
" | tee -a $OUT/sweep.txt
done
