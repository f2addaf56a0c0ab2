#!/bin/bash
# tuning probe: size sweep + PMC counters for one library build
LIB=${1:-gpurun_scratch/libndq_d.so}
TAG=${2:-probe}
export TMPDIR=/tmp
REPO=$(pwd)
mkdir -p gpurun_out/$TAG
for n in 8192 32768 65536 131072 262144 1048576; do KBENCH_N=$n python scripts/kbench.py $LIB 2>/dev/null; done | tee gpurun_out/$TAG/sweep.log
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  KBENCH_N=65536 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $REPO/gpurun_out/$TAG/pmc$i -o pmc -- python $REPO/scripts/kbench.py $REPO/$LIB > $REPO/gpurun_out/$TAG/pmc$i.log 2>&1
done
cd $REPO
find gpurun_out/$TAG -name "*.csv" | head -20
python scripts/pmc_summary.py gpurun_out/$TAG 2>&1 | tail -40
