#!/bin/bash
set -u
TAG=${1:-r04x}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
export GCR_LIB_PATH=$R/tools/_build/libgcr_hip_exp.so
: > $O/${TAG}_k6_cap.jsonl
for rep in 1 2; do
  for pad in 0 4608 10240 18432; do
    echo "{\"k6_lds_pad\": $pad}" >> $O/${TAG}_k6_cap.jsonl
    GCR_K6_LDS_PAD=$pad timeout 300 python bench.py --steps 400 --warmup 50 2>/dev/null >> $O/${TAG}_k6_cap.jsonl
  done
done
echo done
