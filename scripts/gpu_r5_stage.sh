#!/bin/bash
# Round 5: A/B of the in-flight weight staging (csrc/ndq_mlp.h stage_weights, NDQ_STAGE_INFLIGHT) on one box, then the GPU suite.
set -u
TAG=${1:-r05f}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
  echo "--- in-flight staging (default), rep $rep"
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-traffic --no-cold-start > $OUT/bench_inflight_$rep.json 2>$OUT/err_a_$rep.log
  python - $OUT/bench_inflight_$rep.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(f"value={d['value']:.4g} ms_per_step={d['ms_per_step']:.5f} closure_us={d['roofline']['avg_launch_us']:.2f} in_fit={d['in_fit']['ms_per_step']:.5f}")
PY
  echo "--- array-by-array staging (-DNDQ_STAGE_INFLIGHT=0), rep $rep"
  NDQ_JIT_FLAGS="-DNDQ_STAGE_INFLIGHT=0" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-traffic --no-cold-start > $OUT/bench_arrays_$rep.json 2>$OUT/err_b_$rep.log
  python - $OUT/bench_arrays_$rep.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(f"value={d['value']:.4g} ms_per_step={d['ms_per_step']:.5f} closure_us={d['roofline']['avg_launch_us']:.2f} in_fit={d['in_fit']['ms_per_step']:.5f}")
PY
done
echo "--- default windows"
timeout 300 python bench.py --no-cpu-baseline --no-configs --no-traffic --no-cold-start > $OUT/bench_inflight_long.json 2>/dev/null
NDQ_JIT_FLAGS="-DNDQ_STAGE_INFLIGHT=0" timeout 300 python bench.py --no-cpu-baseline --no-configs --no-traffic --no-cold-start > $OUT/bench_arrays_long.json 2>/dev/null
python - $OUT <<'PY'
import json, sys
for n in ("inflight", "arrays"):
    d = json.load(open(f"{sys.argv[1]}/bench_{n}_long.json")); print(n, f"value={d['value']:.4g} ms_per_step={d['ms_per_step']:.5f} closure_us={d['roofline']['avg_launch_us']:.2f}")
PY
timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest_gpu.log | cut -c1-200
du -sh $OUT
