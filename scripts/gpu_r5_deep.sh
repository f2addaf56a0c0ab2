#!/bin/bash
# Round 5: deep wide networks (csrc/ndq_deep.h) -- parity tests, A/B timings of the XCD-aware workgroup ids and of the folded
# head against the round-4 behaviour, kernel trace and HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the default.
#   usage: scripts/gpu_r5_deep.sh [TAG]
set -u
TAG=${1:-r05b}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
timeout 1200 python -m pytest tests/test_gpu_wide.py -q -p no:cacheprovider -x > $OUT/pytest_gpu_wide.log 2>&1; echo "pytest rc=$?"; tail -n 4 $OUT/pytest_gpu_wide.log | cut -c1-300
CFGS=${CFGS:-w18:256 w19:256}
echo "--- default (XCD remap + folded head)"; timeout 300 python scripts/wide_bench.py $CFGS > $OUT/wide_default.jsonl 2> $OUT/wide_default.err; cut -c1-260 $OUT/wide_default.jsonl
echo "--- remap only"; NDQ_JIT_FLAGS="-DNDQ_DEEP_HEAD_FUSED=0" timeout 300 python scripts/wide_bench.py $CFGS > $OUT/wide_remap_only.jsonl 2> $OUT/wide_remap_only.err; cut -c1-260 $OUT/wide_remap_only.jsonl
echo "--- round 4 (no remap, head_bwd pass)"; NDQ_JIT_FLAGS="-DNDQ_DEEP_XCD_REMAP=0 -DNDQ_DEEP_HEAD_FUSED=0" timeout 300 python scripts/wide_bench.py $CFGS > $OUT/wide_r4.jsonl 2> $OUT/wide_r4.err; cut -c1-260 $OUT/wide_r4.jsonl
for cfg in $CFGS; do
  tag=${cfg%%:*}
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$REPO/$OUT/prof_$tag" -o trace -- python "$REPO/scripts/wide_bench.py" $cfg > "$REPO/$OUT/prof_$tag.log" 2>&1)
  python scripts/rocpd_stats.py $OUT/prof_$tag/trace_results.db > $OUT/${tag}_kernel_stats.md 2>/dev/null; head -n 16 $OUT/${tag}_kernel_stats.md | cut -c1-230
  for c in "FETCH_SIZE" "WRITE_SIZE"; do
    (cd "$REPO" && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$REPO/$OUT/pmc_${tag}_$c" -o pmc -- python scripts/wide_bench.py $cfg > "$REPO/$OUT/pmc_${tag}_$c.log" 2>&1)
  done
done
python scripts/pmc_summary.py $OUT > $OUT/pmc_deep_summary.txt 2>/dev/null; head -n 60 $OUT/pmc_deep_summary.txt | cut -c1-220
rm -rf $OUT/pmc_w1*_FETCH_SIZE $OUT/pmc_w1*_WRITE_SIZE $OUT/prof_w18 $OUT/prof_w19
find $OUT -name "*.db" -delete 2>/dev/null
du -sh $OUT
