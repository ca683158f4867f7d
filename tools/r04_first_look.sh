#!/bin/bash
# Round 4, first look (before any change): kernel timeline + host API timeline of the default bench command,
# and the frame loop driven by 2 / 3 host threads.   usage: gpurun --timeout 900 -- 'bash tools/r04_first_look.sh'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --no-secondary"
$B > $O/r04a_bench_c3_before.json 2>$O/r04a_err.txt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o k -- $B > /dev/null 2>&1 < /dev/null
python $R/tools/timeline.py /tmp/kt/k_results.db 600 48 > $O/r04a_timeline_c3_before.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --hip-trace -d /tmp/ht -o k -- $B --steps 100 > /dev/null 2>&1 < /dev/null
python $R/tools/timeline.py /tmp/ht/k_results.db 600 24 > $O/r04a_timeline_c3_before_hip.txt 2>&1
python - <<'PY' > $O/r04a_rocpd_tables.txt 2>&1
import sqlite3
c = sqlite3.connect("/tmp/ht/k_results.db")
for (n, t) in c.execute("select name, type from sqlite_master").fetchall():
    print(t, n)
PY
$B --host-threads 2 > $O/r04a_bench_c3_threads2.json 2>>$O/r04a_err.txt
$B --host-threads 3 > $O/r04a_bench_c3_threads3.json 2>>$O/r04a_err.txt
$B --inference-loop --steps 240 > $O/r04a_bench_inference_loop.json 2>>$O/r04a_err.txt
echo done
