#!/bin/bash
set -u
TAG=${1:-r04g}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --no-secondary"
: > $O/${TAG}_chain.jsonl
run() {
  echo "{\"variant\": \"$*\"}" >> $O/${TAG}_chain.jsonl
  timeout 300 $B "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps({'value':d['value'],'ms':d['ms_per_step'],'lat':d.get('frame_latency_ms'),'other':(d.get('other_entry_point') or {}).get('value'),'stages':{k:[v['ms'],v['ms_single_stream']] for k,v in d.get('stages_ms',{}).items()}}))" >> $O/${TAG}_chain.jsonl
}
for rep in 1 2; do
  run --chain-k1 0
  run --chain-k1 1
  run --chain-k1 0 --steps 20
  run --chain-k1 1 --steps 20
done
run --chain-k1 1 --streams 4
run --chain-k1 1 --streams 2
run --chain-k1 0 --config C5 --steps 48
run --chain-k1 1 --config C5 --steps 48
for c in 0 1; do
  GCR_CHAIN_K1=$c timeout 300 python $R/bench.py --inference-loop --steps 240 --host-camera closed-form 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps({'variant':'inference loop closed-form chain $c','value':d['value']}))" >> $O/${TAG}_chain.jsonl
done
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o k -- $B --chain-k1 1 > /dev/null 2>&1 < /dev/null
python $R/tools/timeline.py /tmp/kt/k_results.db 600 48 > $O/${TAG}_timeline_c3_chain.txt 2>&1
echo done
