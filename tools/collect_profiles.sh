#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun):
#   kernel trace of the default bench command (C3 only, so per-kernel averages are comparable
#   with bench.py's stage timers), then FETCH_SIZE / WRITE_SIZE / SQ counters in SEPARATE --pmc
#   passes (TCC slots do not fit one pass; never combined with sys/hip traces).
# Output: gpurun_out/<tag>_*/  -> summarise with tools/rocpd_stats.py / tools/rocpd_pmc.py
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_kt -o k -- $B --steps 24 > $R/gpurun_out/${TAG}_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/${TAG}_fetch -o p -- $B --steps 8 --warmup 2 > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/${TAG}_write -o p -- $B --steps 8 --warmup 2 > $R/gpurun_out/${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  --kernel-trace -d $R/gpurun_out/${TAG}_sq -o p -- $B --steps 8 --warmup 2 > $R/gpurun_out/${TAG}_sq.log 2>&1
echo collected $TAG
