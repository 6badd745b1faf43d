#!/bin/bash
# usage: scripts/gpu_pmc.sh tag "COUNTER1 COUNTER2 ..." [bench opts]   -> per-kernel counter averages
TAG=$1; PMC=$2; shift; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc -o pmc -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > $OUT/pmc.log 2>&1)
f=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY' | tee $OUT/pmc_summary.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0][:40]
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); 
    cnt[(k, r['Counter_Name'])] += 1
for k in sorted(acc):
    if not k.startswith(('void k_', 'k_')): continue
    print(k, ' '.join(f"{c}={v / cnt[(k, c)]:.3g}" for c, v in sorted(acc[k].items())))
PY
find $OUT/pmc -name "*.csv" -size +5M -delete
