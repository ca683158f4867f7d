#!/bin/bash
# Per-kernel durations of one bench command with every kernel alone on the GPU (--streams 1):
#   gpurun -- 'bash tools/trace_one_stream.sh OUT.txt --config C5 [more bench args]'
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$1; shift
cd /tmp && export TMPDIR=/tmp
d=/tmp/trace_$$
timeout 600 rocprofv3 --kernel-trace --stats -d $d -o k -- python $R/bench.py --no-cpu-baseline --no-secondary --streams 1 "$@" > /dev/null 2>&1 < /dev/null
mkdir -p $(dirname $R/$out)
python $R/tools/rocpd_stats.py $d/k_results.db $R/$out > /dev/null
