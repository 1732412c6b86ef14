#!/bin/bash
# GPU box: SQ counter passes over the default 1024 x 10 s step (tools/x3_step.py, GEMM=<mode>). usage: tools/pmc_step.sh <tag> [mode]
# (counters only: no trace domains in the same rocprofv3 run; TCP_*/TA_* sets abort rocprofv3 on this image)
TAG=${1:-step}
export GEMM=${2:-f16mx}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $line -d $OUT/p$i -o run --output-format csv -- python3 $REPO/tools/x3_step.py > $OUT/p$i.log 2>&1
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU
SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE
LIST
python3 $REPO/tools/pmc_summary.py $OUT tdnn_ > $OUT/summary.txt
python3 $REPO/tools/pmc_summary.py $OUT frontend512 >> $OUT/summary.txt
head -60 $OUT/summary.txt
