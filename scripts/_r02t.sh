OUT=gpurun_out/r02t; mkdir -p $OUT

for v in hip lb2_hip; do
timeout 400 python scripts/ab_bench.py fluidlab_amd/csrc/libfluidengine_$v.so --no-cpu-baseline --steps 8 --warmup 2 > $OUT/bench_$v.json 2>/dev/null
python - <<PY
import json
b=json.loads([l for l in open('gpurun_out/r02t/bench_$v.json') if l.startswith('{')][-1])
print('$v', 'ice', b['extra']['config5_icecream_256_1M']['pairs_per_s'], b['extra']['config5_icecream_256_1M']['kernels_us'])
PY
done
