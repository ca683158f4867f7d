#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o k -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > /dev/null 2>&1 < /dev/null
python $R/tools/burst_timeline.py /tmp/kt/k_results.db 3 > $O/r04l_burst_timeline.txt 2>&1
python $R/tools/burst_timeline.py /tmp/kt/k_results.db 7 > $O/r04l_burst_timeline_b.txt 2>&1
echo done
