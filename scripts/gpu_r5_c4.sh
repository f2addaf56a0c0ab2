#!/bin/bash
# Round 5: C4 (grouped closure kernel) -- A/B of the in-flight staging for multi-output networks + the coordinate prefetch, then
# the GPU suite at the shipping kernels.
set -u
TAG=${1:-r05h}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then export NDQ_JIT_FLAGS="-DNDQ_STAGE_INFLIGHT=0 -DNDQ_GROUP_PREFETCH=0"; else unset NDQ_JIT_FLAGS; fi
    timeout 300 python bench.py --config c4 --steps 50 --warmup 20 > $OUT/c4_${v}_$rep.json 2>$OUT/c4_${v}_$rep.err
    python - $OUT/c4_${v}_$rep.json $v $rep <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(f"c4 {sys.argv[2]} rep {sys.argv[3]}: ms_per_step={d['ms_per_step']:.5f} value={d['value']:.4g} single_launch={d.get('single_launch')}")
PY
  done
done
unset NDQ_JIT_FLAGS
timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest_gpu.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $OUT/smoke.log
du -sh $OUT
