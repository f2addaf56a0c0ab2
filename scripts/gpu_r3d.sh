#!/bin/bash
# round 3, visit D: inverse problems, per-network staging in the multi closure, bench with live traffic
set -u
OUT=gpurun_out/r3d; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
timeout 1500 python -m pytest tests/test_gpu_fit.py tests/test_gpu_dist2.py tests/test_autograd_ops.py -x -q -m gpu -p no:cacheprovider > $OUT/new_tests.log 2>&1; echo "new tests rc=$?"; tail -n 12 $OUT/new_tests.log
timeout 600 python scripts/default_fit.py 6000 > $OUT/default_fit.json 2> $OUT/default_fit.err; echo "default_fit rc=$?"; cat $OUT/default_fit.json; tail -n 3 $OUT/default_fit.err
timeout 600 python scripts/phase_ts.py c1 > $OUT/phase_c1.log 2>&1; echo "phase c1 rc=$?"; grep "stage" $OUT/phase_c1.log | tail -n 3
timeout 900 python bench.py --cold-start > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3d/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "in_fit")}); print(d["roofline"]); print(d.get("cold_start"))
for k, v in d.get("configs", {}).items(): print(k, {a: v[a] for a in ("ms_per_step", "ms_per_step_run_train_epoch", "ms_per_step_in_fit", "frac_of_fp32_mfma_peak", "launches_per_step")})
print(d.get("cpu_baseline", {}).get("value"), d.get("speedup_vs_cpu_baseline"))
PY
tail -n 5 $OUT/bench.err
timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider --deselect tests/test_gpu_fit.py --deselect tests/test_gpu_dist2.py --deselect tests/test_autograd_ops.py > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 6 $OUT/pytest_gpu.log
