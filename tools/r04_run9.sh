#!/bin/bash
set -u
TAG=${1:-r04k}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --no-secondary"
: > $O/${TAG}_pacing.jsonl
run() {
  echo "{\"variant\": \"$*\"}" >> $O/${TAG}_pacing.jsonl
  timeout 300 $B "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps({'value':d['value'],'ms':d['ms_per_step'],'min':d['value_min'],'max':d['value_max'],'other':(d.get('other_entry_point') or {}).get('value')}))" >> $O/${TAG}_pacing.jsonl
}
for rep in 1 2; do
for mu in 0 1 2 3 4; do
  run --max-unresolved $mu --steps 20 --warmup 5
done
done
for mu in 0 1 2 3; do
  run --max-unresolved $mu
done
run --max-unresolved 2 --steps 20 --warmup 5 --streams 2
run --max-unresolved 2 --steps 20 --warmup 5 --streams 4
cd $R; timeout 300 python -m pytest tests/test_gpu_async.py -m gpu -q --timeout 180 2>&1 | tail -3
echo done
