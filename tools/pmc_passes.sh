#!/bin/bash
# PMC passes (one counter set per pass, kernel-trace only) over tools/pmc_conv.py; usage: tools/pmc_passes.sh <outdir>
# Environment switches (B2M_CONV_PIPE=..., ...) set by the caller are inherited by the profiled program.
out=${1:-gpurun_out/pmc}
mkdir -p $out
export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
  "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
  "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python3 tools/pmc_conv.py > $out/p$i.log 2>&1
done
python3 tools/pmc_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
