#!/bin/bash
# instruction-cache and issue counters of the substep kernels on the driver's command.  usage: scripts/gpu_pmc_icache.sh <tag>
TAG=${1:-ic}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python $PWD/bench.py --gpus 1 --steps 20 --warmup 5 --no-probe --no-extras --no-cpu-baseline"
(cd /tmp && rocprofv3 --list-avail 2>&1 | grep -o -E "\b(SQC?_[A-Z0-9_]+)\b" | sort -u > $OUT/counters_sq.txt)
wc -l $OUT/counters_sq.txt
grep -E "ICACHE|IFETCH|WAIT_INST|INSTS_VALU$|INSTS_SALU|BUSY_CYCLES|WAVE_CYCLES|ACTIVE_INST_VALU|INST_CYCLES_VMEM|WAIT_ANY|INSTS_LDS|ACTIVE_INST_LDS|LDS_BANK|INSTS_VMEM" $OUT/counters_sq.txt | tr '\n' ' '
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY"; do
  D=$OUT/pmc_$(echo $C | tr ' ' '_' | cut -c1-40); echo "== pmc $C"
  (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o p -- $CMD > $D.log 2>&1); tail -2 $D.log | cut -c1-200
  F=$(find $D -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python - "$F" <<'P'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
rows = list(csv.DictReader(open(sys.argv[1])))
seen = set()
for r in rows:
    k = r['Kernel_Name'][:34]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r['Dispatch_Id']);
    if key not in seen: seen.add(key); cnt[k] += 1
for k in sorted(acc, key=lambda k: -cnt[k])[:8]:
    extra = ''
    if 'SQ_INSTS_VALU' in acc[k]:        # 1,024 SIMDs, 4 cycles per wave64 VALU instruction, 2.4 GHz
        extra = f" | VALU issue time {acc[k]['SQ_INSTS_VALU'] / cnt[k] * 4 / 1024 / 2400:.1f} us per launch, {acc[k]['SQ_INSTS_VALU'] / cnt[k] / 3125:.0f} VALU instructions per 64 particles"
    print(f'{k:36s} n={cnt[k]:5d} ' + ' '.join(f'{c}={v / cnt[k]:.0f}' for c, v in sorted(acc[k].items())) + extra)
P
  rm -rf $D
done
