#!/bin/bash
# rocprofv3 PMC passes (separate runs, kernel-trace only -- MI355X_MICROARCH.md, HBM / PMC-slot sections) over one command.
# usage: scripts/gpu_pmc.sh TAG [command ...]      default command: the headline bench, short
# The first passes also run scripts/pmc_calib.hip (known byte counts in this package's access patterns) so that
# FETCH_SIZE / WRITE_SIZE can be read against a calibration from the same box and the same tool.
TAG=${1:-pmc}; shift
CMD=("$@")
[ ${#CMD[@]} -eq 0 ] && CMD=(python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-configs)
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$(pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/pmc_calib.hip -o /tmp/pmc_calib || echo "pmc_calib build failed"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$REPO/$OUT/calib_$c" -o pmc -- /tmp/pmc_calib 512 > "$REPO/$OUT/calib_$c.log" 2>&1
  echo "calib $c rc=$?"
done
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-24)
  (cd "$REPO" && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$REPO/$OUT/$n" -o pmc -- "${CMD[@]}" > "$REPO/$OUT/$n.log" 2>&1)
  echo "$c rc=$?"
done
cd "$REPO"
python scripts/pmc_summary.py "$OUT" | tee "$OUT/summary.txt"
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -size +8M -delete
du -sh "$OUT"
