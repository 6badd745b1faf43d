OUT=gpurun_out/r02s; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest_gpu.txt
timeout 600 python bench.py --no-cpu-baseline --steps 20 > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/pytest_gpu.txt; python - <<'PY'
import json
b=json.loads([l for l in open('gpurun_out/r02s/bench.json') if l.startswith('{')][-1])
print('value', b['value'], 'rest', b['extra']['restart_from_rest_pairs_per_s'], '1M', b['extra']['config5_water_256_1M']['pairs_per_s'], 'ice', b['extra']['config5_icecream_256_1M']['pairs_per_s'], b['extra']['config5_icecream_256_1M']['kernels_us'], 'batch', b['extra']['batched_envs']['ratio'])
PY
