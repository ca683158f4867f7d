#!/bin/bash
# SQ counters of the backward blend at C2 (three --pmc passes; rocprofv3 with --kernel-trace only).  usage: gpurun -- 'bash tools/k7_pmc.sh <tag> [piece]'
TAG=${1:-k7}; PIECE=${2:-128}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out; mkdir -p $O
B="python $R/tools/fb_loop.py --piece $PIECE --frames 16"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/${TAG}_$i -o p -- $B > /dev/null 2>&1 < /dev/null
  python $R/tools/pmc_kernel.py /tmp/${TAG}_$i/p_results.db k_blend_bwd
done > $O/${TAG}_k7_pmc.txt
cat $O/${TAG}_k7_pmc.txt
