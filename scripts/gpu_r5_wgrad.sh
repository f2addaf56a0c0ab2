#!/bin/bash
# Round 5: weight gradients of the deep wide networks on the bf16 matrix core (deep_wgrad_bf) -- A/B on one box against the
# exact-f32 kernel (deep_wgrad_gemm) and between occupancy / prefetch-depth settings.  The variants must have been pre-built
# (scripts/prebuild.py under the same NDQ_JIT_FLAGS).   usage: scripts/gpu_r5_wgrad.sh [TAG]
set -u
TAG=${1:-r05m}
OUT=gpurun_out/$TAG; mkdir -p $OUT
CFGS=${CFGS:-w18:256 w19:256 w20:256}
i=0
for f in "" "-DNDQ_DEEP_WGBF_OCC=1 -DNDQ_DEEP_WGBF_PF=4" "-DNDQ_DEEP_WGBF_OCC=1 -DNDQ_DEEP_WGBF_PF=2" "-DNDQ_DEEP_WGRAD_BF=0"; do
  echo "--- flags: '$f'"
  NDQ_JIT_FLAGS="$f" timeout 300 python scripts/wide_bench.py $CFGS 2> $OUT/v$i.err | tee -a $OUT/wgrad_ab.jsonl | cut -c1-120
  i=$((i+1))
done
