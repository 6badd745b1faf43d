#!/bin/bash
# rocprofv3 of the 1M-particle block (scripts/prof_1m.py): kernel stats, HBM bytes and VALU instructions per launch -> gpurun_out/<tag>/r04_1m_*.txt
TAG=${1:-m1}; export M1TAG=$TAG; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python $PWD/scripts/prof_1m.py 4 water"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1)
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_256_1M_water.csv; rm -rf $OUT/trace; grep n_used $OUT/trace.log
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES"; do      # (FETCH_SIZE and WRITE_SIZE in passes of their own, as MI355X_MICROARCH.md prescribes)
  D=$OUT/pmc; (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o p -- $CMD > $D.log 2>&1)
  F=$(find $D -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python - "$F" <<'P' | tee -a $OUT/pmc_256_1M_water.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:34]; acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in seen: seen.add(r['Dispatch_Id']); cnt[k] += 1
for k in sorted(acc, key=lambda k: -cnt[k])[:10]:
    a = acc[k]; extra = ''
    if 'SQ_INSTS_VALU' in a: extra = f" | VALU issue time {a['SQ_INSTS_VALU'] / cnt[k] * 4 / 1024 / 2400:.1f} us per launch, {a['SQ_INSTS_VALU'] / cnt[k] / 15625:.0f} VALU instructions per 64 particles"
    if 'FETCH_SIZE' in a: extra = f" | fetch {2 * a['FETCH_SIZE'] / cnt[k] / 1024:.1f} MB per launch (KB counter x 2: the gfx950 correction of MI355X_MICROARCH.md for 16 B/lane reads)"
    if 'WRITE_SIZE' in a: extra = f" | write {a['WRITE_SIZE'] / cnt[k] / 1024:.1f} MB per launch (KB counter)"
    print(f'{k:36s} n={cnt[k]:5d} ' + ' '.join(f'{c}={v / cnt[k]:.0f}' for c, v in sorted(a.items())) + extra)
P
  rm -rf $D
done
python - <<'P'
import csv
for r in csv.reader(open('gpurun_out/'+__import__('os').environ.get('M1TAG','m1')+'/kernel_stats_256_1M_water.csv')):
    if r[0] != 'Name' and float(r[4]) > 0.5: print(r[0][:40], r[1], round(float(r[3]) / 1000, 1), 'us', r[4], '%')
P
