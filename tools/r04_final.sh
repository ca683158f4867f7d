#!/bin/bash
# Final pass of the round after the last kernel change (gcr_blend.hip): counters + traces of the rasterizer path again,
# the bench lines that quote them, the N = 2 control flow of bench.py on one GPU (gloo), the GPU test suite.
set -u
TAG=r04
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --timeout 180 > $O/${TAG}_pytest_final.txt 2>&1
tail -3 $O/${TAG}_pytest_final.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-secondary"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_kt -o k -- $B > /dev/null 2>&1 < /dev/null
python $R/tools/rocpd_stats.py /tmp/${TAG}_kt/k_results.db $O/${TAG}_kernel_trace_stats.txt > /dev/null
python $R/tools/timeline.py /tmp/${TAG}_kt/k_results.db 600 48 > $O/${TAG}_timeline_c3.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_kt1 -o k -- $B --streams 1 > /dev/null 2>&1 < /dev/null
python $R/tools/rocpd_stats.py /tmp/${TAG}_kt1/k_results.db $O/${TAG}_kernel_trace_stats_one_stream.txt > /dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/${TAG}_f -o p -- $B --steps 8 --warmup 2 > /dev/null 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/${TAG}_w -o p -- $B --steps 8 --warmup 2 > /dev/null 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  --kernel-trace -d /tmp/${TAG}_sq -o p -- $B --steps 8 --warmup 2 > /dev/null 2>&1 < /dev/null
python $R/tools/rocpd_pmc.py /tmp/${TAG}_f/p_results.db $O/${TAG}_pmc_fetch.txt > /dev/null
python $R/tools/rocpd_pmc.py /tmp/${TAG}_w/p_results.db $O/${TAG}_pmc_write.txt > /dev/null
python $R/tools/rocpd_pmc.py /tmp/${TAG}_sq/p_results.db $O/${TAG}_pmc_sq.txt > /dev/null
python $R/tools/make_traffic.py /tmp/${TAG}_f/p_results.db /tmp/${TAG}_w/p_results.db $O/${TAG}_traffic.json /tmp/${TAG}_sq/p_results.db > /dev/null
B="python $R/bench.py --config C2 --backward --no-cpu-baseline --no-secondary"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_c2_kt -o k -- $B --steps 200 --streams 1 > /dev/null 2>&1 < /dev/null
python $R/tools/rocpd_stats.py /tmp/${TAG}_c2_kt/k_results.db $O/${TAG}_c2_bwd_kernel_trace_stats.txt > /dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/${TAG}_c2_f -o p -- $B --steps 8 --warmup 2 > /dev/null 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/${TAG}_c2_w -o p -- $B --steps 8 --warmup 2 > /dev/null 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  --kernel-trace -d /tmp/${TAG}_c2_sq -o p -- $B --steps 8 --warmup 2 > /dev/null 2>&1 < /dev/null
python $R/tools/rocpd_pmc.py /tmp/${TAG}_c2_f/p_results.db $O/${TAG}_c2_bwd_pmc_fetch.txt > /dev/null
python $R/tools/rocpd_pmc.py /tmp/${TAG}_c2_w/p_results.db $O/${TAG}_c2_bwd_pmc_write.txt > /dev/null
python $R/tools/rocpd_pmc.py /tmp/${TAG}_c2_sq/p_results.db $O/${TAG}_c2_bwd_pmc_sq.txt > /dev/null
python $R/tools/make_traffic.py /tmp/${TAG}_c2_f/p_results.db /tmp/${TAG}_c2_w/p_results.db $O/${TAG}_traffic_c2.json \
  /tmp/${TAG}_c2_sq/p_results.db "C2 (500k S-rand, 640x448, SH3, forward + backward)" > /dev/null
# the counters just taken are what the bench lines below quote (bench.py reads the newest profiles/rNN_traffic*.json)
cp $O/${TAG}_traffic.json $O/${TAG}_traffic_c2.json $R/profiles/
python $R/bench.py > $O/${TAG}_bench_c3.json 2>/dev/null
python $R/bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_c3_driver_like.json 2>/dev/null
python $R/bench.py --config C2 --backward --no-secondary --steps 200 > $O/${TAG}_bench_c2_fwd_bwd.json 2>/dev/null
# N = 2 control flow on ONE GPU (gloo; not a measurement): sharding, rank report, max-over-ranks, one JSON line
GCR_BENCH_SHARE_GPU=1 timeout 600 python $R/bench.py --gpus 2 --steps 20 --no-cpu-baseline --no-secondary > $O/${TAG}_n2_control_flow_one_gpu.json 2>$O/${TAG}_n2_err.txt
GCR_BENCH_SHARE_GPU=1 timeout 600 python $R/bench.py --gpus 2 --train-step --steps 50 > $O/${TAG}_n2_trainstep_control_flow_one_gpu.json 2>>$O/${TAG}_n2_err.txt
tail -3 $O/${TAG}_n2_err.txt
echo done
