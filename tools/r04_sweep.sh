#!/bin/bash
# C3 headline A/Bs in the GPU-bound regime (frames without the host wait): stream counts, split K1, sort in blend
set -u
TAG=${1:-r04d}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --no-secondary"
: > $O/${TAG}_sweep.jsonl
for v in "" "--streams 2" "--streams 4" "--streams 5" "--streams 6" "--split-preprocess" "--split-preprocess --streams 4" "--sort-in-blend" "--sort-in-blend --streams 4" "--host-threads 2" "" ; do
  echo "{\"variant\": \"$v\"}" >> $O/${TAG}_sweep.jsonl
  timeout 200 $B $v 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps({'value':d['value'],'ms':d['ms_per_step'],'min':d['value_min'],'max':d['value_max'],'other':d.get('other_entry_point',{}).get('value'),'stages':{k:[v['ms'],v['ms_single_stream']] for k,v in d['stages_ms'].items()}}))" >> $O/${TAG}_sweep.jsonl
done
echo done
