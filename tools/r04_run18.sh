#!/bin/bash
set -u
TAG=${1:-r04v}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
: > $O/${TAG}_ab.jsonl
for rep in 1 2; do
  for v in ship static rows4; do
    echo "{\"variant\": \"K6 $v\"}" >> $O/${TAG}_ab.jsonl
    if [ $v = ship ]; then timeout 300 python bench.py --steps 400 --warmup 50 2>/dev/null >> $O/${TAG}_ab.jsonl
    else GCR_LIB_PATH=$R/tools/_build/libgcr_hip_$v.so timeout 300 python bench.py --steps 400 --warmup 50 2>/dev/null >> $O/${TAG}_ab.jsonl; fi
  done
done
timeout 300 python tools/k6_clocks.py > $O/${TAG}_k6_clocks.jsonl 2>/dev/null
cat $O/${TAG}_k6_clocks.jsonl
echo done
