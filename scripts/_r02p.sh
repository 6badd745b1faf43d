OUT=gpurun_out/r02p; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bench.py -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest_bench.txt
timeout 600 python bench.py --gpus 1 --replicas --steps 2 --warmup 1 > $OUT/replica_c3.json 2> $OUT/replica_c3.err
timeout 300 python scripts/ubench_enqueue.py > $OUT/enqueue.txt 2>&1
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o trace -- python $OLDPWD/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > $OLDPWD/$OUT/rocprof_bench.log 2>&1)
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); ls -la $f; python scripts/trace_breakdown.py $f 10 > $OUT/trace_breakdown.txt 2>&1
tail -3 $OUT/pytest_bench.txt; tail -c 1500 $OUT/replica_c3.json; tail -5 $OUT/replica_c3.err; cat $OUT/enqueue.txt | tail -3; head -50 $OUT/trace_breakdown.txt
