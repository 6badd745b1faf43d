OUT=gpurun_out/s2; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_random_scenes.py -m gpu -x -q 2>&1 | tail -2
bash scripts/gpu_trace_short.sh tr 20 5 > $OUT/trace_short.txt 2>&1
python - <<'P'
import csv
tot=0
for r in csv.reader(open('gpurun_out/tr/profiles/tr_kernel_stats_timed_region.csv')):
    if r[0] != 'Name':
        tot+=float(r[2])
        if 'sort' in r[0] or 'perm' in r[0]: print(r[0][:30], r[1], r[3], r[5], r[6])
print('kernel us per pair', tot/2e6)
P
grep -o '"value": [0-9.]*' gpurun_out/tr/trace.log
