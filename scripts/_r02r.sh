OUT=gpurun_out/r02r; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest_gpu.txt
timeout 600 python bench.py --no-cpu-baseline --steps 100 > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python scripts/evolve_probe.py 104 16 > $OUT/evolve.txt 2>&1
tail -5 $OUT/pytest_gpu.txt; python - <<'PY'
import json
b=json.loads([l for l in open('gpurun_out/r02r/bench.json') if l.startswith('{')][-1])
print('value', b['value'], 'rest', b['extra']['restart_from_rest_pairs_per_s'], '1M', b['extra']['config5_water_256_1M']['pairs_per_s'], b['extra']['config5_water_256_1M']['kernels_us'], 'ice', b['extra']['config5_icecream_256_1M']['pairs_per_s'], 'batch', b['extra']['batched_envs']['ratio'])
print({k: v['avg_us'] for k, v in b['kernels'].items()})
PY
cut -c1-420 $OUT/evolve.txt
timeout 300 python scripts/latteart_probe.py config3 40 2>&1 | tail -1 | cut -c1-200
