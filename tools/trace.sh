#!/bin/bash
# kernel timeline of a bench.py command line ($1 = tag, rest = bench args)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$TAG -o k -- python $R/bench.py --no-cpu-baseline --no-secondary "$@" > /tmp/kt_$TAG.log 2>&1 < /dev/null
tail -1 /tmp/kt_$TAG.log | cut -c1-300
python $R/tools/rocpd_stats.py /tmp/kt_$TAG/k_results.db $O/${TAG}_kernel_trace_stats.txt > /dev/null
python $R/tools/timeline.py /tmp/kt_$TAG/k_results.db 600 60 > $O/${TAG}_timeline.txt 2>&1
tail -8 $O/${TAG}_timeline.txt
