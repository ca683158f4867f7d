#!/bin/bash
# SQ / LDS counters of the forward blend at C3, shipped library vs a variant (three --pmc passes each; rocprofv3 with
# --kernel-trace only).  usage: gpurun -- 'bash tools/k6_pmc.sh <tag> [variant ...]'
TAG=${1:-k6}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out; mkdir -p $O
B="python $R/bench.py --steps 8 --warmup 2"
: > $O/${TAG}_k6_pmc.txt
for v in ship "$@"; do
  echo "== $v" >> $O/${TAG}_k6_pmc.txt
  if [ $v = ship ]; then unset GCR_LIB_PATH; else export GCR_LIB_PATH=$R/tools/_build/libgcr_hip_$v.so; fi
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/${TAG}_${v}_$i -o p -- $B > /dev/null 2>&1 < /dev/null
    python $R/tools/pmc_kernel.py /tmp/${TAG}_${v}_$i/p_results.db k_blend_fwd >> $O/${TAG}_k6_pmc.txt
  done
done
cat $O/${TAG}_k6_pmc.txt
