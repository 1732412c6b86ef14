#!/bin/bash
# on the GPU box: LDS / VALU counters of the front-end launch for each given library build (tools/fe_time.py under rocprofv3 --pmc)
REPO=$(pwd)
mkdir -p $REPO/gpurun_out/fe_pmc
cd /tmp && export TMPDIR=/tmp
export KTF_ALLOW_LIBRARY_OVERRIDE=1
for lib in "$@"; do
  n=$(basename $lib .so)
  export KTF_LIBRARY=$REPO/$lib
  rm -rf /tmp/fepmc_$n
  timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/fepmc_$n -o run --output-format csv -- python3 $REPO/tools/fe_time.py > /tmp/fepmc_$n.log 2>&1
  echo "== $n" | tee -a $REPO/gpurun_out/fe_pmc/summary.txt
  python3 $REPO/tools/pmc_summary.py /tmp/fepmc_$n "frontend512_kernel<false" | head -9 | tee -a $REPO/gpurun_out/fe_pmc/summary.txt
done
