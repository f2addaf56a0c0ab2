#!/bin/bash
# round 3, second evidence run (after the transposing-read weight gradients): GPU suite, smoke, bench line (+ driver
# protocol), kernel trace of the bench, default-size fit() figures, PMC passes for the C2 and C3 closure kernels.
# usage: scripts/gpu_r3zz.sh [TAG]
set -u
TAG=${1:-r03zz}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
(rocminfo | grep -E "Marketing|gfx" | head -4; nproc; lscpu | grep "Model name") > $OUT/env.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 4 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-500 $OUT/bench.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $OUT/bench_driver_protocol.json 2>/dev/null; cut -c1-420 $OUT/bench_driver_protocol.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/$OUT/prof" -o trace -- python "$REPO/bench.py" --steps 2000 --warmup 200 --no-cpu-baseline --no-traffic --no-cold-start > "$REPO/$OUT/prof_bench.json" 2> "$REPO/$OUT/prof.err"); echo "rocprof rc=$?"
python scripts/rocpd_stats.py $OUT/prof/trace_results.db > $OUT/bench_kernel_stats.md 2>/dev/null; head -n 8 $OUT/bench_kernel_stats.md | cut -c1-220
timeout 600 python scripts/default_fit.py 6000 > $OUT/default_fit.json 2> $OUT/default_fit.err; echo "default_fit rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/default_fit.json")); print({k: v for k, v in d.items() if k.endswith("_us_per_epoch") and "host" not in k or "identical" in k})
PY
bash scripts/gpu_pmc.sh ${TAG}_pmc_c2 > $OUT/pmc_c2.log 2>&1; tail -n 12 $OUT/pmc_c2.log
bash scripts/gpu_pmc.sh ${TAG}_pmc_c3 python bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-traffic --no-cold-start > $OUT/pmc_c3.log 2>&1; tail -n 12 $OUT/pmc_c3.log
find $OUT gpurun_out/${TAG}_pmc_c2 gpurun_out/${TAG}_pmc_c3 -name "*.db" -size +30M -delete
du -sh gpurun_out/${TAG}*
