#!/bin/bash
# round 3 evidence: GPU suite, smoke, bench line (+ driver protocol), kernel trace of the bench, default-size fit() figures
# and their kernel traces, PMC passes for the headline kernel.   usage: scripts/gpu_r3z.sh [TAG]
set -u
TAG=${1:-r03z}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
(rocminfo | grep -E "Marketing|gfx" | head -4; nproc; lscpu | grep "Model name") > $OUT/env.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 4 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-500 $OUT/bench.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $OUT/bench_driver_protocol.json 2>/dev/null; cut -c1-420 $OUT/bench_driver_protocol.json
ROC_ACTIVE_WAIT_TIMEOUT=100000 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-traffic > $OUT/bench_driver_protocol_active_wait.json 2>/dev/null; cut -c1-420 $OUT/bench_driver_protocol_active_wait.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/$OUT/prof" -o trace -- python "$REPO/bench.py" --steps 2000 --warmup 200 --no-cpu-baseline --no-traffic > "$REPO/$OUT/prof_bench.json" 2> "$REPO/$OUT/prof.err"); echo "rocprof rc=$?"
python scripts/rocpd_stats.py $OUT/prof/trace_results.db > $OUT/bench_kernel_stats.md 2>/dev/null; head -n 8 $OUT/bench_kernel_stats.md | cut -c1-220
timeout 600 python scripts/default_fit.py 6000 > $OUT/default_fit.json 2> $OUT/default_fit.err; echo "default_fit rc=$?"
NDQ_FIT_PULL=0 timeout 600 python scripts/default_fit.py 6000 > $OUT/default_fit_two_launches.json 2> /dev/null; echo "default_fit (two launches per epoch) rc=$?"
python - <<PY
import json
for f in ("default_fit", "default_fit_two_launches"):
    d = json.load(open("$OUT/%s.json" % f)); print(f, {k: v for k, v in d.items() if k.endswith("_us_per_epoch") and "host" not in k or "identical" in k})
PY
for p in ode pde system; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$REPO/$OUT/prof_fit_$p" -o trace -- python "$REPO/scripts/fit_profile.py" $p 3000 > "$REPO/$OUT/prof_fit_$p.log" 2>&1); grep "us/epoch" $OUT/prof_fit_$p.log
  python scripts/rocpd_stats.py $OUT/prof_fit_$p/trace_results.db > $OUT/fit_${p}_kernel_stats.md 2>/dev/null; sed -n 3,4p $OUT/fit_${p}_kernel_stats.md | cut -c1-200
done
NDQ_JIT_FLAGS=-DNDQ_PHASE_TS timeout 600 python scripts/pull_ts.py 300 > $OUT/pull_ts.log 2>&1; tail -n 4 $OUT/pull_ts.log
bash scripts/gpu_pmc.sh ${TAG}_pmc_c2 > $OUT/pmc_c2.log 2>&1; tail -n 12 $OUT/pmc_c2.log
find $OUT -name "*.db" -size +30M -delete
du -sh gpurun_out/${TAG}*
