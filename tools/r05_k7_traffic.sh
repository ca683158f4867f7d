#!/bin/bash
set -u
TAG=${1:-r05m}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out; mkdir -p $O
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
head -8 $O/${TAG}_c2_bwd_kernel_trace_stats.txt | cut -c1-130
python - <<PY
import json
d=json.load(open("$O/${TAG}_traffic_c2.json"))
for k in ("blend_fwd","blend_bwd","preprocess_bwd"):
    v=d["kernels"][k]; print(k, "FETCH KB", round(v["FETCH_SIZE_KB"]), "WRITE KB", round(v["WRITE_SIZE_KB"]), "traffic MB", round((2*v["FETCH_SIZE_KB"]+v["WRITE_SIZE_KB"])*1.024/1000,1), "VALU", v["SQ_INSTS_VALU"])
PY
