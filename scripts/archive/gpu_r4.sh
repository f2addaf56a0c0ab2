#!/bin/bash
# Round 4 evidence on one MI355X box: GPU suite, smoke, bench (default + driver protocol), kernel traces, the wide / deep
# network timings with their traces and PMC counters, the multi-GPU dry run.   usage: scripts/gpu_r4.sh [TAG] [what...]
#   what: tests bench trace wide pmc dryrun   (default: all)
set -u
TAG=${1:-r04z}; shift || true
WHAT=${*:-tests bench trace wide pmc dryrun}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
has() { [[ " $WHAT " == *" $1 "* ]]; }
(rocminfo | grep -E "Marketing|gfx" | head -4; nproc; lscpu | grep "Model name") > $OUT/env.log 2>&1
if has tests; then
  timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 5 $OUT/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $OUT/smoke.log
fi
if has bench; then
  timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-600 $OUT/bench.json
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $OUT/bench_driver_protocol.json 2>/dev/null; cut -c1-420 $OUT/bench_driver_protocol.json
fi
if has trace; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/$OUT/prof" -o trace -- python "$REPO/bench.py" --steps 2000 --warmup 200 --no-cpu-baseline --no-traffic > "$REPO/$OUT/prof_bench.json" 2> "$REPO/$OUT/prof.err"); echo "rocprof rc=$?"
  python scripts/rocpd_stats.py $OUT/prof/trace_results.db > $OUT/bench_kernel_stats.md 2>/dev/null; head -n 8 $OUT/bench_kernel_stats.md | cut -c1-220
  python scripts/step_timeline.py $OUT/prof/trace_results.db > $OUT/step_timeline.json 2>$OUT/step_timeline.err; cat $OUT/step_timeline.json
fi
if has wide; then
  timeout 600 python scripts/wide_bench.py w16:256 w17:256 w18:256 w19:256 w20:256 w16:1024 > $OUT/wide.jsonl 2> $OUT/wide.err; cat $OUT/wide.jsonl | cut -c1-330
  for cfg in w16:256 w18:256 w19:256; do
    tag=${cfg%%:*}
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$REPO/$OUT/prof_$tag" -o trace -- python "$REPO/scripts/wide_bench.py" $cfg > "$REPO/$OUT/prof_$tag.log" 2>&1)
    python scripts/rocpd_stats.py $OUT/prof_$tag/trace_results.db > $OUT/${tag}_kernel_stats.md 2>/dev/null; head -n 14 $OUT/${tag}_kernel_stats.md | cut -c1-200
  done
fi
if has pmc; then
  # SQ counters of the wide closure kernel (VALU-bound by design: no MFMA) and of the deep GEMMs, separate passes
  for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
    n=$(echo $c | tr ' ' '_' | cut -c1-40)
    for cfg in w16:256 w18:256; do
      tag=${cfg%%:*}
      (cd "$REPO" && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$REPO/$OUT/pmc_${tag}_$n" -o pmc -- python scripts/wide_bench.py $cfg > "$REPO/$OUT/pmc_${tag}_$n.log" 2>&1)
    done
  done
  python scripts/pmc_summary.py $OUT > $OUT/pmc_wide_summary.txt 2>/dev/null; head -n 40 $OUT/pmc_wide_summary.txt | cut -c1-200
  rm -rf $OUT/pmc_w16_* $OUT/pmc_w18_*            # (the raw counter csv files are hundreds of MB; the summary is what is kept)
  bash scripts/gpu_pmc.sh ${TAG}_pmc_c2 > $OUT/pmc_c2.log 2>&1; tail -n 12 $OUT/pmc_c2.log
fi
if has dryrun; then
  bash scripts/scale_dryrun.sh $OUT/scale_dryrun
fi
find $OUT gpurun_out/${TAG}_pmc_c2 -name "*.db" -delete 2>/dev/null
rm -rf $OUT/prof $OUT/prof_w16 $OUT/prof_w18 $OUT/prof_w19
find $OUT gpurun_out/${TAG}_pmc_c2 -name "*.csv" -size +2M -delete 2>/dev/null
du -sh $OUT
