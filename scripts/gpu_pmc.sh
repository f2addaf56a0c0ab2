#!/bin/bash
# HBM traffic counters for the bench kernels (separate --pmc passes, kernel-trace only; MI355X_MICROARCH.md HBM section)
TAG=${1:-pmc}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-24)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$REPO/$OUT/$n" -o pmc -- python "$REPO/bench.py" --steps 30 --warmup 5 --no-cpu-baseline > "$REPO/$OUT/$n.log" 2>&1
  echo "$c rc=$?"
done
cd "$REPO"
python scripts/pmc_summary.py "$OUT" | tee "$OUT/summary.txt"
find "$OUT" -name "*kernel_trace.csv" -delete
du -sh "$OUT"
