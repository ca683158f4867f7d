#!/bin/bash
# round 6 first look: gpu tests, default bench lines, one-stream trace + timeline (baseline of the round on one box)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r06_first_pytest.txt 2>&1; tail -3 $O/r06_first_pytest.txt
python bench.py > $O/r06_first_bench_c3.json 2>/dev/null; tail -1 $O/r06_first_bench_c3.json | cut -c1-600
python bench.py --steps 20 --no-cpu-baseline --no-secondary > $O/r06_first_bench_c3_k20.json 2>/dev/null
python bench.py --config C5 --steps 48 --no-secondary --no-cpu-baseline > $O/r06_first_bench_c5.json 2>/dev/null
python bench.py --config C2 --backward --no-secondary --no-cpu-baseline --steps 200 > $O/r06_first_bench_c2.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-secondary"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o k -- $B > /dev/null 2>&1 < /dev/null
python $R/tools/rocpd_stats.py /tmp/kt/k_results.db $O/r06_first_kernel_trace_stats.txt > /dev/null
python $R/tools/timeline.py /tmp/kt/k_results.db 600 48 > $O/r06_first_timeline_c3.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt1 -o k -- $B --streams 1 > /dev/null 2>&1 < /dev/null
python $R/tools/rocpd_stats.py /tmp/kt1/k_results.db $O/r06_first_kernel_trace_stats_one_stream.txt > /dev/null
B="python $R/bench.py --config C5 --no-cpu-baseline --no-secondary --steps 48"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt5 -o k -- $B --streams 1 > /dev/null 2>&1 < /dev/null
python $R/tools/rocpd_stats.py /tmp/kt5/k_results.db $O/r06_first_kernel_trace_stats_one_stream_c5.txt > /dev/null
echo done
