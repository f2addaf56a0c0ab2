#!/bin/bash
# round 3, visit A: multi-epoch fit path
set -u
OUT=gpurun_out/r3a; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fit.py -x -q -p no:cacheprovider > $OUT/fit_tests.log 2>&1; echo "fit tests rc=$?"; tail -n 25 $OUT/fit_tests.log
timeout 600 python scripts/default_fit.py 3000 > $OUT/default_fit.json 2> $OUT/default_fit.err; echo "default_fit rc=$?"; cat $OUT/default_fit.json; tail -n 5 $OUT/default_fit.err
timeout 300 python bench.py --no-cpu-baseline --no-configs > $OUT/bench_fit1.json 2> $OUT/bench_fit1.err; echo "bench(fit_run) rc=$?"; cut -c1-500 $OUT/bench_fit1.json
NDQ_FIT_RUN=0 timeout 300 python bench.py --no-cpu-baseline --no-configs > $OUT/bench_fit0.json 2> $OUT/bench_fit0.err; echo "bench(old path) rc=$?"; cut -c1-500 $OUT/bench_fit0.json
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --deselect tests/test_gpu_fit.py > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 12 $OUT/pytest_gpu.log
