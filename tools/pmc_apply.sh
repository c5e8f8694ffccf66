#!/bin/bash
# SQ counter passes over the operator-apply kernels (tools/apply_only.py); one rocprofv3 --pmc run per group (8 SQ slots per pass).
# usage (on the GPU box): bash tools/pmc_apply.sh <tag> [resolution] [degree]      -> gpurun_out/pmc_<tag>_<pass>.csv
tag=${1:-x}; r=${2:-2}; deg=${3:-1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
i=0
for group in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_LDS_ADDR_CONFLICT" \
             "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i+1))
  rocprofv3 --pmc $group --output-format csv -d gpurun_out/pmc_${tag}_$i -- python3 tools/apply_only.py $r 3 $deg > gpurun_out/pmc_${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pmc_${tag}_$i.log; }
  f=$(find gpurun_out/pmc_${tag}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py "$f" > gpurun_out/pmc_${tag}_$i.txt && cat gpurun_out/pmc_${tag}_$i.txt
  rm -rf gpurun_out/pmc_${tag}_$i
done
