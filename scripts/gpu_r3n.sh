#!/bin/bash
# round 3, visit N: where a tile's cycles go with the transposing-read weight gradients (Cfg::WG_TR) and without
set -u
OUT=gpurun_out/r3n; mkdir -p $OUT
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 scripts/ubench_tr16.hip -o /tmp/ubench_tr16 2>/dev/null && /tmp/ubench_tr16 > $OUT/tr16.log 2>&1; echo "tr16 rc=$?"; tail -n 3 $OUT/tr16.log
for v in "tr:-DNDQ_PHASE_TS" "f32:-DNDQ_PHASE_TS -DNDQ_WG_TR=0"; do
  tag=${v%%:*}; flags=${v#*:}
  NDQ_JIT_FLAGS="$flags" timeout 300 python scripts/phase_ts.py c2 > $OUT/ts_${tag}_8w.log 2>&1; echo "== $tag 8-wave"; tail -n 22 $OUT/ts_${tag}_8w.log
  NDQ_FUSED_WIDE_MIN=1000000000 NDQ_JIT_FLAGS="$flags" timeout 300 python scripts/phase_ts.py c2 > $OUT/ts_${tag}_4w.log 2>&1; echo "== $tag 4-wave"; tail -n 22 $OUT/ts_${tag}_4w.log
done
