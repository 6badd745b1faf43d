OUT=gpurun_out/r02u; mkdir -p $OUT
timeout 500 python -m pytest tests -m gpu -x -q -k "parity or fullsize or golden or config" 2>&1 | tail -3
for v in prev_hip hip; do
timeout 400 python scripts/ab_bench.py fluidlab_amd/csrc/libfluidengine_$v.so --no-cpu-baseline > $OUT/bench_$v.json 2>/dev/null
python - <<PY
import json
b=json.loads([l for l in open('gpurun_out/r02u/bench_$v.json') if l.startswith('{')][-1])
print('$v', 'value', b['value'], 'rest', b['extra']['restart_from_rest_pairs_per_s'], '1M', b['extra']['config5_water_256_1M']['pairs_per_s'], 'ice', b['extra']['config5_icecream_256_1M']['pairs_per_s'], {k: v['avg_us'] for k, v in b['kernels'].items()})
PY
done
