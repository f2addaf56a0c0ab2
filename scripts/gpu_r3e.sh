#!/bin/bash
# round 3, visit E: pull mode (one launch per epoch for small grids)
set -u
OUT=gpurun_out/r3e; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
timeout 1500 python -m pytest tests/test_gpu_fit.py -x -q -m gpu -p no:cacheprovider > $OUT/fit_tests.log 2>&1; echo "fit tests rc=$?"; tail -n 15 $OUT/fit_tests.log
timeout 600 python scripts/default_fit.py 6000 > $OUT/default_fit.json 2> $OUT/default_fit.err; echo "default_fit rc=$?"; cat $OUT/default_fit.json; tail -n 3 $OUT/default_fit.err
NDQ_FIT_PULL=0 timeout 600 python scripts/default_fit.py 6000 > $OUT/default_fit_nopull.json 2> /dev/null; echo "default_fit (two launches per epoch) rc=$?"; python -c "
import json; d=json.load(open('$OUT/default_fit_nopull.json')); print({k:v for k,v in d.items() if k.endswith('fit_us_per_epoch')})"
for p in ode pde system; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$REPO/$OUT/prof_$p" -o trace -- python "$REPO/scripts/fit_profile.py" $p 3000 > "$REPO/$OUT/prof_$p.log" 2>&1); grep "us/epoch" $OUT/prof_$p.log
  python scripts/rocpd_stats.py $OUT/prof_$p/trace_results.db 2>/dev/null | cut -c1-200 | head -5
done
timeout 600 python scripts/bench_configs.py c1 c4 > $OUT/bench_c1_c4.json 2> $OUT/bench_c1_c4.err; cat $OUT/bench_c1_c4.json | cut -c1-300
NDQ_GROUP_WIDE=1 timeout 900 python scripts/bench_configs.py c4 > $OUT/bench_c4_wide.json 2> $OUT/bench_c4_wide.err; echo "c4 8-wave rc=$?"; cut -c1-300 $OUT/bench_c4_wide.json; tail -n 3 $OUT/bench_c4_wide.err
timeout 900 python bench.py --no-cpu-baseline --no-traffic > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3e/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "in_fit")}); print(d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
for k, v in d.get("configs", {}).items(): print(k, {a: v[a] for a in ("ms_per_step", "ms_per_step_run_train_epoch", "ms_per_step_in_fit", "frac_of_fp32_mfma_peak")})
PY
timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider --deselect tests/test_gpu_fit.py > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 6 $OUT/pytest_gpu.log
