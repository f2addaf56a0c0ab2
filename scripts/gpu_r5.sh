#!/bin/bash
# Round 5 evidence on one MI355X box.   usage: scripts/gpu_r5.sh [TAG] [what...]
#   what: tests smoke bench benchq trace wide pmc dryrun   (default: tests smoke benchq)
set -u
TAG=${1:-r05a}; shift || true
WHAT=${*:-tests smoke benchq}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
has() { [[ " $WHAT " == *" $1 "* ]]; }
(rocminfo | grep -E "Marketing|gfx" | head -4; nproc; lscpu | grep "Model name") > $OUT/env.log 2>&1
if has tests; then
  timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider ${PYTEST_ARGS:-} > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 12 $OUT/pytest_gpu.log | cut -c1-300
fi
if has smoke; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $OUT/smoke.log
fi
if has benchq; then
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $OUT/bench_driver_protocol.json 2>$OUT/benchq.err; echo "benchq rc=$?"; tail -c 1500 $OUT/bench_driver_protocol.json
fi
if has bench; then
  timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 2500 $OUT/bench.json
fi
if has trace; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/$OUT/prof" -o trace -- python "$REPO/bench.py" --steps 2000 --warmup 200 --no-cpu-baseline --no-traffic > "$REPO/$OUT/prof_bench.json" 2> "$REPO/$OUT/prof.err"); echo "rocprof rc=$?"
  python scripts/rocpd_stats.py $OUT/prof/trace_results.db > $OUT/bench_kernel_stats.md 2>/dev/null; head -n 10 $OUT/bench_kernel_stats.md | cut -c1-260
  python scripts/step_timeline.py $OUT/prof/trace_results.db > $OUT/step_timeline.json 2>$OUT/step_timeline.err; cat $OUT/step_timeline.json
fi
if has wide; then
  timeout 600 python scripts/wide_bench.py ${WIDE_CFGS:-w16:256 w17:256 w18:256 w19:256} > $OUT/wide.jsonl 2> $OUT/wide.err; cat $OUT/wide.jsonl | cut -c1-330
  for cfg in ${WIDE_TRACE:-w18:256}; do
    tag=${cfg%%:*}
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$REPO/$OUT/prof_$tag" -o trace -- python "$REPO/scripts/wide_bench.py" $cfg > "$REPO/$OUT/prof_$tag.log" 2>&1)
    python scripts/rocpd_stats.py $OUT/prof_$tag/trace_results.db > $OUT/${tag}_kernel_stats.md 2>/dev/null; head -n 16 $OUT/${tag}_kernel_stats.md | cut -c1-220
  done
fi
if has pmc; then
  bash scripts/gpu_pmc.sh ${TAG}_pmc_c2 > $OUT/pmc_c2.log 2>&1; tail -n 12 $OUT/pmc_c2.log
fi
if has dryrun; then
  bash scripts/scale_dryrun.sh $OUT/scale_dryrun
fi
if has custom; then
  bash -c "${CUSTOM_CMD}" > $OUT/custom.log 2>&1; echo "custom rc=$?"; tail -n 40 $OUT/custom.log | cut -c1-300
fi
find $OUT -name "*.db" -delete 2>/dev/null
rm -rf $OUT/prof $OUT/prof_w16 $OUT/prof_w18 $OUT/prof_w19
find $OUT -name "*.csv" -size +2M -delete 2>/dev/null
du -sh $OUT
