#!/bin/bash
# round 3, visit G: pull mode with every prologue load in flight at once
set -u
OUT=gpurun_out/r3h; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
timeout 1200 python -m pytest tests/test_gpu_fit.py -x -q -m gpu -p no:cacheprovider > $OUT/fit_tests.log 2>&1; echo "fit tests rc=$?"; tail -n 5 $OUT/fit_tests.log
timeout 600 python scripts/default_fit.py 6000 > $OUT/default_fit.json 2> $OUT/default_fit.err; echo "default_fit rc=$?"; tail -n 3 $OUT/default_fit.err
NDQ_FIT_PULL=0 timeout 600 python scripts/default_fit.py 6000 > $OUT/default_fit_nopull.json 2> /dev/null; echo "default_fit (two launches per epoch) rc=$?"
python - <<'PY'
import json
for f in ("default_fit", "default_fit_nopull"):
    d = json.load(open(f"gpurun_out/r3h/{f}.json")); print(f, {k: v for k, v in d.items() if k.endswith("fit_us_per_epoch") or "identical" in k})
PY
for p in ode pde system; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$REPO/$OUT/prof_$p" -o trace -- python "$REPO/scripts/fit_profile.py" $p 3000 > "$REPO/$OUT/prof_$p.log" 2>&1); grep "us/epoch" $OUT/prof_$p.log
  python scripts/rocpd_stats.py $OUT/prof_$p/trace_results.db 2>/dev/null | cut -c1-200 | sed -n 3,4p
done
for mode in 1 0; do
  NDQ_FIT_PULL=$mode timeout 900 python bench.py --no-cpu-baseline --no-traffic > $OUT/bench_pull$mode.json 2> $OUT/bench_pull$mode.err; echo "bench pull=$mode rc=$?"
  python - <<PY
import json
d = json.loads(open("$OUT/bench_pull$mode.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["in_fit"]["ms_per_step"])
for k, v in d.get("configs", {}).items(): print(k, {a: v[a] for a in ("ms_per_step_run_train_epoch", "ms_per_step_in_fit")})
PY
done
NDQ_JIT_FLAGS=-DNDQ_PHASE_TS timeout 600 python scripts/pull_ts.py 300 > $OUT/pull_ts.log 2>&1; tail -n 8 $OUT/pull_ts.log
