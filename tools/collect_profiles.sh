#!/bin/bash
# Collects the round's rocprofv3 evidence ON THE GPU BOX (run through gpurun) and writes only the text / JSON
# summaries into gpurun_out/ (the rocpd databases stay in /tmp: gpurun_out/ is capped at 64 MiB):
#   kernel trace of the default bench command, then FETCH_SIZE / WRITE_SIZE / SQ counters in SEPARATE --pmc passes
#   (TCC slots do not fit one pass; never combined with sys/hip traces), for the three bench paths.
# Afterwards copy gpurun_out/<tag>_* into profiles/ and commit.
#   usage: gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01'
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O

# ---- rasterizer (headline path) ----------------------------------------------------------------------------
B="python $R/bench.py --no-cpu-baseline --no-secondary"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_kt -o k -- $B > /dev/null 2>&1 < /dev/null
python $R/tools/rocpd_stats.py /tmp/${TAG}_kt/k_results.db $O/${TAG}_kernel_trace_stats.txt > /dev/null
# the frame period, K1-end -> next-K1-start gaps, queue busy fractions: the default command (frames
# without the host wait), the positional int API (one host wait per frame), and a host API timeline of the default command
python $R/tools/timeline.py /tmp/${TAG}_kt/k_results.db 600 48 > $O/${TAG}_timeline_c3.txt 2>&1
timeout 600 rocprofv3 --kernel-trace -d /tmp/${TAG}_kti -o k -- $B --native-int-api > /dev/null 2>&1 < /dev/null
python $R/tools/timeline.py /tmp/${TAG}_kti/k_results.db 600 48 > $O/${TAG}_timeline_c3_native_int_api.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --hip-trace -d /tmp/${TAG}_ht -o k -- $B --steps 100 > /dev/null 2>&1 < /dev/null
python $R/tools/timeline.py /tmp/${TAG}_ht/k_results.db 600 18 2>&1 | head -260 > $O/${TAG}_timeline_c3_hip_trace.txt
# the same command on ONE stream: every kernel alone on the GPU -- the durations bench.py's `roofline.launch_ms` and
# `stages_ms[*].ms_single_stream` report (the default command's trace above shows them stretched by the frames in flight)
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

# ---- a static scene's second line (bench.py --static-scene: K1 reads the cull cache): one-stream trace + counters ----------
B="python $R/bench.py --static-scene --no-cpu-baseline --no-secondary"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_ss_kt1 -o k -- $B --streams 1 > /dev/null 2>&1 < /dev/null
python $R/tools/rocpd_stats.py /tmp/${TAG}_ss_kt1/k_results.db $O/${TAG}_kernel_trace_stats_one_stream_static_scene.txt > /dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/${TAG}_ss_f -o p -- $B --steps 8 --warmup 2 > /dev/null 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/${TAG}_ss_w -o p -- $B --steps 8 --warmup 2 > /dev/null 2>&1 < /dev/null
python $R/tools/make_traffic.py /tmp/${TAG}_ss_f/p_results.db /tmp/${TAG}_ss_w/p_results.db $O/${TAG}_traffic_static_scene.json \
  "" "C3 (5M S-city, 1920x1080, SH3, forward) with the static scene's cull cache" > /dev/null

# ---- C2 forward + backward (BASELINE metric's second half): kernel trace + counters for K7 / K8 ---------------
B="python $R/bench.py --config C2 --backward --no-cpu-baseline --no-secondary"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_c2_kt -o k -- $B --steps 200 > /dev/null 2>&1 < /dev/null
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
# ---- C5 (20 M Gaussians, 4K): counters for its roofline.traffic -------------------------------------------
B="python $R/bench.py --config C5 --no-cpu-baseline --no-secondary"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/${TAG}_c5_f -o p -- $B --steps 8 --warmup 2 > /dev/null 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/${TAG}_c5_w -o p -- $B --steps 8 --warmup 2 > /dev/null 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  --kernel-trace -d /tmp/${TAG}_c5_sq -o p -- $B --steps 8 --warmup 2 > /dev/null 2>&1 < /dev/null
python $R/tools/make_traffic.py /tmp/${TAG}_c5_f/p_results.db /tmp/${TAG}_c5_w/p_results.db $O/${TAG}_traffic_c5.json \
  /tmp/${TAG}_c5_sq/p_results.db "C5 (20M S-city, 3840x2160, SH3, forward)" > /dev/null

# ---- C4: the wrapper / helpers path at the product's own shape (host-bound): kernel trace + host profile ---------------
B="python $R/bench.py --train-step --steps 200 --host-camera closed-form"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_c4_kt -o k -- $B > /dev/null 2>&1 < /dev/null
python $R/tools/rocpd_stats.py /tmp/${TAG}_c4_kt/k_results.db $O/${TAG}_c4_kernel_trace_stats.txt > /dev/null
(cd $R && timeout 200 python tools/host_profile_c4.py --host-camera > $O/${TAG}_c4_host_profile.txt 2>/dev/null)
(cd $R && timeout 200 python tools/host_profile_c4.py --device-camera > $O/${TAG}_c4_host_profile_device_camera.txt 2>/dev/null)
python $R/bench.py --train-step --steps 200 --host-camera device > $O/${TAG}_bench_c4_trainstep_device_camera.json 2>/dev/null
python $R/bench.py --train-step --steps 200 > $O/${TAG}_bench_c4_trainstep.json 2>/dev/null
python $R/bench.py --train-step --steps 200 --host-camera closed-form > $O/${TAG}_bench_c4_trainstep_host_camera.json 2>/dev/null
python $R/bench.py --train-step --steps 200 --host-camera reference > $O/${TAG}_bench_c4_trainstep_host_camera_reference.json 2>/dev/null
[ -n "${RASTER_ONLY:-}" ] && { python $R/bench.py > $O/${TAG}_bench_c3.json 2>/dev/null; echo collected $TAG raster only; ls $O | grep "^${TAG}_"; exit 0; }

# ---- visibility and hash-grid encoder (SKIP_F2F3=1: rows f2 / f3 unchanged this round, only their bench lines) ----------
for P in visibility grid-encoder; do
  [ -n "${SKIP_F2F3:-}" ] && continue
  N=${P/-/_}
  B="python $R/bench.py --path $P --no-cpu-baseline"
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_${N}_kt -o k -- $B --steps 24 --warmup 3 > /dev/null 2>&1 < /dev/null
  python $R/tools/rocpd_stats.py /tmp/${TAG}_${N}_kt/k_results.db $O/${TAG}_${N}_kernel_trace_stats.txt > /dev/null
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/${TAG}_${N}_f -o p -- $B --steps 6 --warmup 2 > /dev/null 2>&1 < /dev/null
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/${TAG}_${N}_w -o p -- $B --steps 6 --warmup 2 > /dev/null 2>&1 < /dev/null
  python $R/tools/rocpd_pmc.py /tmp/${TAG}_${N}_f/p_results.db $O/${TAG}_${N}_pmc_fetch.txt > /dev/null
  python $R/tools/rocpd_pmc.py /tmp/${TAG}_${N}_w/p_results.db $O/${TAG}_${N}_pmc_write.txt > /dev/null
done

# ---- bench lines ---------------------------------------------------------------------------------------------------
python $R/bench.py > $O/${TAG}_bench_c3.json 2>/dev/null
python $R/bench.py --native-int-api --no-cpu-baseline --no-secondary > $O/${TAG}_bench_c3_native_int_api.json 2>/dev/null
python $R/bench.py --config C5 --steps 48 --no-secondary --no-cpu-baseline > $O/${TAG}_bench_c5.json 2>/dev/null
python $R/bench.py --config D1 --steps 48 --no-secondary --no-cpu-baseline > $O/${TAG}_bench_d1.json 2>/dev/null   # dense stress scene (lazy tile sort)
python $R/bench.py --config D1 --steps 48 --no-secondary --no-cpu-baseline --sort-whole > $O/${TAG}_bench_d1_sorted_whole.json 2>/dev/null
python $R/bench.py --path visibility > $O/${TAG}_bench_visibility.json 2>/dev/null
python $R/bench.py --path grid-encoder > $O/${TAG}_bench_grid_encoder.json 2>/dev/null
for hc in "--host-camera device" "--host-camera reference" "--host-camera closed-form" "--host-camera reference --float-frames" "--host-camera closed-form --float-frames"; do   # the product's inference loop through the wrapper
  python $R/bench.py --inference-loop --steps 240 $hc 2>/dev/null
done > $O/${TAG}_bench_inference_loop.jsonl
python $R/bench.py --config C2 --backward --no-secondary --steps 200 > $O/${TAG}_bench_c2_fwd_bwd.json 2>/dev/null
python $R/bench.py --config C2 --backward --no-secondary --no-cpu-baseline --steps 200 --bwd-wave-units > $O/${TAG}_bench_c2_fwd_bwd_wave_units.json 2>/dev/null   # round 4's backward blend kernel (A/B)
python $R/bench.py --steps 20 --no-cpu-baseline --no-secondary > $O/${TAG}_bench_c3_k20.json 2>/dev/null
for c in C3 C5; do for m in "" "--native-int-api"; do   # the static scene's second line, stateless beside it on the same box
  python $R/bench.py --config $c $m --no-cpu-baseline --no-secondary 2>/dev/null | tail -1
  python $R/bench.py --config $c $m --static-scene --no-cpu-baseline --no-secondary 2>/dev/null | tail -1
done; done > $O/${TAG}_bench_static_scene.jsonl
echo collected $TAG; ls $O | grep "^${TAG}_"
