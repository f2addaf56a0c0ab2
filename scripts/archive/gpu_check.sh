#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprofv3 kernel trace.  Everything lands under gpurun_out/.
# usage: scripts/gpu_check.sh [tag]
set -u
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
REPO=$(pwd)
echo "== rocminfo ==" > "$OUT/env.log"; (rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -8; nproc; lscpu | grep "Model name") >> "$OUT/env.log" 2>&1
echo "== pytest -m gpu ==" 
timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest_gpu.log"
tail -n 15 "$OUT/pytest_gpu.log"
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/smoke.log"; tail -n 3 "$OUT/smoke.log"
echo "== bench =="
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cat "$OUT/bench.json"; tail -n 5 "$OUT/bench.err"
echo "== rocprofv3 kernel trace =="
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/$OUT/prof" -o trace -- python "$REPO/bench.py" --steps 2000 --warmup 200 --no-cpu-baseline > "$REPO/$OUT/prof_bench.json" 2> "$REPO/$OUT/prof.err"); echo "rocprof rc=$?"
find "$OUT/prof" -name "*stats*" | head; f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -n 25 "$f"
# keep the merge-back small: drop the raw per-dispatch trace if it is large
find "$OUT/prof" -name "*kernel_trace.csv" -size +20M -delete
du -sh "$OUT"
echo "== bench under torch.distributed.run (1 rank, RCCL path) =="
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5000 --warmup 500 > "$OUT/bench_dist1.json" 2> "$OUT/bench_dist1.err"; echo "dist bench rc=$?"; tail -n 1 "$OUT/bench_dist1.json" | cut -c1-400; tail -n 3 "$OUT/bench_dist1.err"
