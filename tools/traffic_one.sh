#!/bin/bash
# Counted HBM traffic + SQ counters of one bench command (separate --pmc passes, as tools/collect_profiles.sh does):
#   gpurun -- 'bash tools/traffic_one.sh OUT.json "workload text" --config C5 [more bench args]'
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$1; shift; what=$1; shift
cd /tmp && export TMPDIR=/tmp
d=/tmp/traffic_$$
B="python $R/bench.py --no-cpu-baseline --no-secondary --steps 8 --warmup 2 $*"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d ${d}_f -o p -- $B > /dev/null 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d ${d}_w -o p -- $B > /dev/null 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  --kernel-trace -d ${d}_sq -o p -- $B > /dev/null 2>&1 < /dev/null
mkdir -p $(dirname $R/$out)
python $R/tools/make_traffic.py ${d}_f/p_results.db ${d}_w/p_results.db $R/$out ${d}_sq/p_results.db "$what" > /dev/null
