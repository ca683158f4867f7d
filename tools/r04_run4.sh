#!/bin/bash
set -u
TAG=${1:-r04f}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_async.py -m gpu -q --timeout 180 > $O/${TAG}_pytest.txt 2>&1
tail -3 $O/${TAG}_pytest.txt
: > $O/${TAG}_affinity_ab.jsonl
for rep in 1 2; do
for mode in none early late; do
  case $mode in none) E="GCR_NO_AFFINITY=1";; early) E="GCR_X=1";; late) E="GCR_AFFINITY=late";; esac
  env $E timeout 300 python bench.py --train-step --steps 200 --host-camera closed-form 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps({'mode':'$mode','what':'c4 train step','step_ms':d['ms_per_step'],'leg_ms':d['rasterizer_leg']['ms_per_frame'],'leg_fastest':d['rasterizer_leg']['ms_per_frame_fastest_block'],'affinity':d['ranks'][0]['affinity']}))" >> $O/${TAG}_affinity_ab.jsonl
  env $E timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps({'mode':'$mode','what':'c3','value':d['value'],'other':d['other_entry_point']['value']}))" >> $O/${TAG}_affinity_ab.jsonl
done
done
GCR_FILL_BLOCKS=0 timeout 300 python tools/k7_knockout.py 128 > $O/${TAG}_k7_no_fill.jsonl 2>/dev/null
timeout 300 python tools/k7_knockout.py 128 > $O/${TAG}_k7_with_fill.jsonl 2>/dev/null
echo done
