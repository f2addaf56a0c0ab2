#!/bin/bash
# H = 64 kernels (C3 / C5): parity subset, throughput, kernel trace and one SQ counter pass of the C3 step.
TAG=${1:-wide}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$(pwd)
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "c3 or c5 or heat_wide" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -n 5 "$OUT/pytest.log"
timeout 300 python scripts/bench_configs.py c3 c5 | tee "$OUT/bench.jsonl"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$REPO/$OUT/prof" -o trace -- python "$REPO/scripts/bench_configs.py" c3 > "$REPO/$OUT/prof.log" 2>&1; echo "trace rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$REPO/$OUT/pmc_sq" -o pmc -- python "$REPO/scripts/bench_configs.py" c3 > "$REPO/$OUT/pmc_sq.log" 2>&1; echo "pmc rc=$?"
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_IFETCH --kernel-trace --output-format csv -d "$REPO/$OUT/pmc_lds" -o pmc -- python "$REPO/scripts/bench_configs.py" c3 > "$REPO/$OUT/pmc_lds.log" 2>&1; echo "pmc2 rc=$?"
cd "$REPO"
python scripts/rocpd_stats.py "$OUT/prof/trace_results.db" 2>/dev/null | head -8
python scripts/pmc_summary.py "$OUT" 2>/dev/null | head -60
find "$OUT" -name "*kernel_trace.csv" -delete
