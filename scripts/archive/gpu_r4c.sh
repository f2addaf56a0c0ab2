#!/bin/bash
# Round 4, supplementary evidence: kernel trace of an fp64 epoch, PMC counters of C3 / C4 and of the deep-network GEMMs.
set -u
TAG=${1:-r04y}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$REPO/$OUT/prof_fp64" -o trace -- python "$REPO/scripts/fp64_step.py" 300 > "$REPO/$OUT/fp64_step.json" 2> "$REPO/$OUT/fp64.err"); cat $OUT/fp64_step.json
python scripts/rocpd_stats.py $OUT/prof_fp64/trace_results.db > $OUT/fp64_kernel_stats.md 2>/dev/null; head -n 14 $OUT/fp64_kernel_stats.md | cut -c1-200
rm -rf $OUT/prof_fp64
bash scripts/gpu_pmc.sh ${TAG}_pmc_c34 python scripts/bench_configs.py c3 c4 > $OUT/pmc_c34.log 2>&1; grep -E "grid/wg|FETCH_SIZE|WRITE_SIZE|MFMA|INSTS_VALU|BUSY_CYCLES" gpurun_out/${TAG}_pmc_c34/summary.txt | grep -A9 "fused_" | head -40
bash scripts/gpu_pmc.sh ${TAG}_pmc_w18 python scripts/wide_bench.py w18:256 > $OUT/pmc_w18.log 2>&1; grep -A16 "deep_gemm_bf\|deep_wgrad" gpurun_out/${TAG}_pmc_w18/summary.txt | head -120
find gpurun_out/${TAG}_pmc_c34 gpurun_out/${TAG}_pmc_w18 -name "*.csv" -delete 2>/dev/null
find gpurun_out/${TAG}_pmc_c34 gpurun_out/${TAG}_pmc_w18 -name "*.db" -delete 2>/dev/null
du -sh gpurun_out/${TAG}*
