#!/bin/bash
# (TCP_*/TA_* counter sets abort rocprofv3 on this image; SQ sets only.)
# Runs on the GPU box: PMC passes over tools/gemm_layers.py (isolated TDNN GEMM layers). usage: tools/pmc_gemm.sh <tag>
TAG=${1:-gemm}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $line -d $OUT/p$i -o run --output-format csv -- python3 $REPO/tools/gemm_layers.py > $OUT/p$i.log 2>&1
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES
LIST
python3 $REPO/tools/pmc_summary.py $OUT tdnn_
