#!/bin/bash
# round 3, visit C: near-convergence parity, C5 at size, DP failure tests, phase timestamps of the small-N closures
set -u
OUT=gpurun_out/r3c; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
export NDQ_SELF_CHECK_LOG=$REPO/$OUT/self_checks.jsonl
timeout 1500 python -m pytest tests/test_gpu_fit.py tests/test_gpu_dist2.py tests/test_autograd_ops.py -x -q -m gpu -p no:cacheprovider > $OUT/new_tests.log 2>&1; echo "new tests rc=$?"; tail -n 12 $OUT/new_tests.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider -k "near_convergence or (at_size and c5)" --durations=5 > $OUT/parity_new.log 2>&1; echo "parity new rc=$?"; tail -n 25 $OUT/parity_new.log
for p in pde ode; do NDQ_FIT_TRACE=1 timeout 300 python scripts/fit_profile.py $p 3000 > $OUT/fit_trace_$p.log 2>&1; head -n 20 $OUT/fit_trace_$p.log; done
timeout 600 python scripts/phase_ts.py c1 > $OUT/phase_c1.log 2>&1; echo "phase c1 rc=$?"; tail -n 25 $OUT/phase_c1.log
timeout 600 python scripts/phase_ts.py c2:2 > $OUT/phase_c2_tiny.log 2>&1; echo "phase c2 tiny rc=$?"; tail -n 25 $OUT/phase_c2_tiny.log
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --deselect tests/test_gpu_fit.py --deselect tests/test_gpu_dist2.py --deselect tests/test_autograd_ops.py -k "not near_convergence and not (at_size and c5)" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 6 $OUT/pytest_gpu.log
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r3c/self_checks.jsonl")]
g = sorted(r["grad_rel_l2"] for r in rows); l = sorted(r["loss_rel"] for r in rows)
print("self checks:", len(rows), "grad rel-L2 median/max", g[len(g)//2], g[-1], "loss rel median/max", l[len(l)//2], l[-1],
      "tv identical:", all(r["train_valid_launch_identical"] for r in rows), "reproducible:", all(r["reproducible"] for r in rows))
PY
