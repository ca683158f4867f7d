#!/bin/bash
set -u
TAG=${1:-r04h}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --timeout 180 -x > $O/${TAG}_pytest.txt 2>&1
tail -4 $O/${TAG}_pytest.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-secondary"
: > $O/${TAG}_bucket.jsonl
run() {
  echo "{\"variant\": \"$*\"}" >> $O/${TAG}_bucket.jsonl
  timeout 300 $B "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps({'value':d['value'],'ms':d['ms_per_step'],'lat':d.get('frame_latency_ms'),'other':(d.get('other_entry_point') or {}).get('value'),'stages':{k:[v['ms'],v['ms_single_stream']] for k,v in d.get('stages_ms',{}).items()}}))" >> $O/${TAG}_bucket.jsonl
}
for rep in 1 2; do
  run --bucket-scatter 0
  run --bucket-scatter 1
done
run --bucket-scatter 0 --config C5 --steps 48
run --bucket-scatter 1 --config C5 --steps 48
run --bucket-scatter 0 --config D1 --steps 48
run --bucket-scatter 1 --config D1 --steps 48
run --bucket-scatter 0 --config C2 --backward --steps 200
run --bucket-scatter 1 --config C2 --backward --steps 200
echo done
