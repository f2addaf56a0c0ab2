#!/bin/bash
# round 3, visit B: concurrent multi-network closure, bulk-draw fix, ADVICE fixes
set -u
OUT=gpurun_out/r3b; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
timeout 900 python -m pytest tests/test_gpu_fit.py tests/test_gpu_dist2.py tests/test_autograd_ops.py -x -q -m gpu -p no:cacheprovider > $OUT/new_tests.log 2>&1; echo "new tests rc=$?"; tail -n 25 $OUT/new_tests.log
timeout 600 python scripts/default_fit.py 3000 > $OUT/default_fit.json 2> $OUT/default_fit.err; echo "default_fit rc=$?"; cat $OUT/default_fit.json; tail -n 5 $OUT/default_fit.err
for p in ode pde system; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$REPO/$OUT/prof_$p" -o trace -- python "$REPO/scripts/fit_profile.py" $p 3000 > "$REPO/$OUT/prof_$p.log" 2>&1); tail -n 1 $OUT/prof_$p.log
  f=$(find $OUT/prof_$p -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -n 6 "$f" | cut -c1-220
  find $OUT/prof_$p -name "*kernel_trace.csv" -size +20M -delete
done
timeout 600 python scripts/bench_configs.py c1 > $OUT/bench_c1.json 2> $OUT/bench_c1.err; echo "bench c1 rc=$?"; cut -c1-600 $OUT/bench_c1.json; tail -n 3 $OUT/bench_c1.err
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --deselect tests/test_gpu_fit.py --deselect tests/test_gpu_dist2.py --deselect tests/test_autograd_ops.py > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 12 $OUT/pytest_gpu.log
