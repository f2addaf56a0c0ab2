#!/bin/bash
# round 4: kernel trace of the deep wide kernels (csrc/ndq_deep.h) on w18 (128 x 3) at 65 536 points
mkdir -p gpurun_out/r04b
REPO=$(pwd); export TMPDIR=/tmp
for cfg in ${1:-w18:256}; do
  tag=${cfg%%:*}
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/r04b/prof_$tag" -o trace -- python "$REPO/scripts/wide_bench.py" $cfg > "$REPO/gpurun_out/r04b/prof_$tag.log" 2>&1); tail -n 1 gpurun_out/r04b/prof_$tag.log
  python scripts/rocpd_stats.py gpurun_out/r04b/prof_$tag/trace_results.db > gpurun_out/r04b/${tag}_kernel_stats.md 2>/dev/null; head -n 24 gpurun_out/r04b/${tag}_kernel_stats.md | cut -c1-250
done
find gpurun_out/r04b -name "*.db" -size +30M -delete
