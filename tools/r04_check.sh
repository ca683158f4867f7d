#!/bin/bash
# Round 4 check run: GPU tests, smoke, the headline bench through both entry points, the product's own loops.
#   usage: gpurun --timeout 1500 -- 'bash tools/r04_check.sh [tag]'
set -u
TAG=${1:-r04b}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --timeout 180 > $O/${TAG}_pytest.txt 2>&1
tail -5 $O/${TAG}_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.txt 2>&1
tail -3 $O/${TAG}_smoke.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
timeout 600 $B > $O/${TAG}_bench_c3.json 2>$O/${TAG}_err.txt
timeout 300 $B --no-cpu-baseline --no-secondary --steps 20 > $O/${TAG}_bench_c3_k20.json 2>>$O/${TAG}_err.txt
timeout 300 $B --inference-loop --steps 240 > $O/${TAG}_bench_inference_loop.json 2>>$O/${TAG}_err.txt
timeout 300 $B --inference-loop --steps 240 --host-camera closed-form > $O/${TAG}_bench_inference_loop_closed_form.json 2>>$O/${TAG}_err.txt
timeout 300 $B --train-step --steps 200 > $O/${TAG}_bench_c4_trainstep.json 2>>$O/${TAG}_err.txt
timeout 300 $B --train-step --steps 200 --host-camera closed-form > $O/${TAG}_bench_c4_trainstep_closed_form.json 2>>$O/${TAG}_err.txt
timeout 300 $B --config C5 --steps 48 --no-secondary --no-cpu-baseline > $O/${TAG}_bench_c5.json 2>>$O/${TAG}_err.txt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o k -- $B --no-cpu-baseline --no-secondary > /dev/null 2>&1 < /dev/null
python $R/tools/timeline.py /tmp/kt/k_results.db 600 48 > $O/${TAG}_timeline_c3.txt 2>&1
python $R/tools/rocpd_stats.py /tmp/kt/k_results.db $O/${TAG}_kernel_trace_stats.txt > /dev/null
echo done
