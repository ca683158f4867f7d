#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out; mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --no-secondary --split-preprocess"
timeout 600 rocprofv3 --kernel-trace -d /tmp/sp_kt -o k -- $B > /dev/null 2>&1 < /dev/null
python $R/tools/timeline.py /tmp/sp_kt/k_results.db 600 60 > $O/r05_timeline_c3_split_k1.txt 2>&1
head -64 $O/r05_timeline_c3_split_k1.txt; tail -8 $O/r05_timeline_c3_split_k1.txt
$B 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('split', d['value'], {k:(v['ms'],v['ms_single_stream']) for k,v in d['stages_ms'].items()})"
