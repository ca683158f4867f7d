#!/bin/bash
set -u
TAG=${1:-r04r}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 180 -x > $O/${TAG}_pytest.txt 2>&1
tail -5 $O/${TAG}_pytest.txt
: > $O/${TAG}_rows_ab.jsonl
for rep in 1 2 3; do
  for v in ship rows4; do
    echo "{\"variant\": \"K6 $v\"}" >> $O/${TAG}_rows_ab.jsonl
    if [ $v = ship ]; then timeout 300 python bench.py --steps 400 --warmup 50 2>/dev/null >> $O/${TAG}_rows_ab.jsonl
    else GCR_LIB_PATH=$R/tools/_build/libgcr_hip_$v.so timeout 300 python bench.py --steps 400 --warmup 50 2>/dev/null >> $O/${TAG}_rows_ab.jsonl; fi
  done
done
for v in ship rows4; do
  echo "{\"variant\": \"K6 $v\"}" >> $O/${TAG}_rows_ab.jsonl
  if [ $v = ship ]; then timeout 300 python tools/piece_probe.py --pieces 128 2>/dev/null >> $O/${TAG}_rows_ab.jsonl
  else GCR_LIB_PATH=$R/tools/_build/libgcr_hip_$v.so timeout 300 python tools/piece_probe.py --pieces 128 2>/dev/null >> $O/${TAG}_rows_ab.jsonl; fi
done
echo done
