#!/bin/bash
# round 3, visit L: loop mode (a run of epochs = one launch of one workgroup) for one-workgroup problems
set -u
OUT=gpurun_out/r3l; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
timeout 900 python -m pytest tests/test_gpu_fit.py -x -q -m gpu -p no:cacheprovider > $OUT/fit_tests.log 2>&1; echo "fit tests rc=$?"; tail -n 12 $OUT/fit_tests.log
timeout 600 python scripts/default_fit.py 6000 > $OUT/default_fit.json 2> $OUT/default_fit.err; echo "default_fit rc=$?"; tail -n 3 $OUT/default_fit.err
NDQ_FIT_LOOP=0 timeout 600 python scripts/default_fit.py 6000 > $OUT/default_fit_noloop.json 2> /dev/null; echo "default_fit (no loop mode) rc=$?"
python - <<'PY'
import json
for f in ("default_fit", "default_fit_noloop"):
    d = json.load(open(f"gpurun_out/r3l/{f}.json")); print(f, {k: v for k, v in d.items() if k.endswith("fit_us_per_epoch") or "identical" in k})
PY
for p in ode system; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$REPO/$OUT/prof_$p" -o trace -- python "$REPO/scripts/fit_profile.py" $p 3000 > "$REPO/$OUT/prof_$p.log" 2>&1); grep "us/epoch" $OUT/prof_$p.log
  python scripts/rocpd_stats.py $OUT/prof_$p/trace_results.db 2>/dev/null | cut -c1-200 | sed -n 3,6p
done
