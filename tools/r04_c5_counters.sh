#!/bin/bash
# C5's counter file and the C5 / D1 bench lines that quote it, alone (tools/collect_profiles.sh does the whole set).
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out; mkdir -p $O
B="python $R/bench.py --config C5 --no-cpu-baseline --no-secondary"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/${TAG}_c5_f -o p -- $B --steps 8 --warmup 2 > /dev/null 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/${TAG}_c5_w -o p -- $B --steps 8 --warmup 2 > /dev/null 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  --kernel-trace -d /tmp/${TAG}_c5_sq -o p -- $B --steps 8 --warmup 2 > /dev/null 2>&1 < /dev/null
python $R/tools/make_traffic.py /tmp/${TAG}_c5_f/p_results.db /tmp/${TAG}_c5_w/p_results.db $O/${TAG}_traffic_c5.json \
  /tmp/${TAG}_c5_sq/p_results.db "C5 (20M S-city, 3840x2160, SH3, forward)" > /dev/null
cp $O/${TAG}_traffic_c5.json $R/profiles/${TAG}_traffic_c5.json
cd $R
python bench.py --config C5 --steps 48 --no-secondary --no-cpu-baseline > $O/${TAG}_bench_c5.json 2>/dev/null
python bench.py --config D1 --steps 48 --no-secondary --no-cpu-baseline > $O/${TAG}_bench_d1.json 2>/dev/null
echo done
