#!/bin/bash
set -u
TAG=${1:-r04o}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
: > $O/${TAG}_micro_ab.jsonl
for rep in 1 2 3; do
  for v in ship k7nosent; do
    echo "{\"variant\": \"K7 $v\"}" >> $O/${TAG}_micro_ab.jsonl
    if [ $v = ship ]; then timeout 300 python tools/piece_probe.py --pieces 128 --no-c3 2>/dev/null >> $O/${TAG}_micro_ab.jsonl
    else GCR_LIB_PATH=$R/tools/_build/libgcr_hip_$v.so timeout 300 python tools/piece_probe.py --pieces 128 --no-c3 2>/dev/null >> $O/${TAG}_micro_ab.jsonl; fi
  done
done
cd /tmp
for rep in 1 2; do
  for v in ship k6noskip; do
    echo "{\"variant\": \"K6 $v\"}" >> $O/${TAG}_micro_ab.jsonl
    if [ $v = ship ]; then L=""; else L="GCR_LIB_PATH=$R/tools/_build/libgcr_hip_$v.so"; fi
    env $L timeout 300 python $R/bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps({'value':d['value'],'blend_alone_ms':d['stages_ms']['blend_fwd']['ms_single_stream'],'blend_in_flight_ms':d['stages_ms']['blend_fwd']['ms']}))" >> $O/${TAG}_micro_ab.jsonl
  done
done
echo done
