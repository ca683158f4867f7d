#!/bin/bash
set -u
TAG=${1:-r04i}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --timeout 180 > $O/${TAG}_pytest.txt 2>&1
tail -4 $O/${TAG}_pytest.txt
: > $O/${TAG}_inference.jsonl
for v in "--host-camera reference" "--host-camera reference --float-frames" "--host-camera closed-form" "--host-camera closed-form --float-frames" "--host-camera reference" "--host-camera closed-form"; do
  timeout 300 python bench.py --inference-loop --steps 240 $v 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps({'variant':'$v','value':d['value'],'min':d['value_min'],'max':d['value_max']}))" >> $O/${TAG}_inference.jsonl
done
timeout 300 python tools/c4_leg_ab.py --camera reference > $O/${TAG}_c4_leg_ab_reference.json 2>/dev/null
echo done
