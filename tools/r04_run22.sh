#!/bin/bash
set -u
TAG=${1:-r04z}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
: > $O/${TAG}_k7_occ.jsonl
for rep in 1 2 3; do
  for v in ship k7w5 k7w6; do
    echo "{\"variant\": \"K7 $v\"}" >> $O/${TAG}_k7_occ.jsonl
    if [ $v = ship ]; then timeout 300 python tools/piece_probe.py --pieces 128 --no-c3 2>/dev/null >> $O/${TAG}_k7_occ.jsonl
    else GCR_LIB_PATH=$R/tools/_build/libgcr_hip_$v.so timeout 300 python tools/piece_probe.py --pieces 128 --no-c3 2>/dev/null >> $O/${TAG}_k7_occ.jsonl; fi
  done
done
echo done
