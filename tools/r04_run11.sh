#!/bin/bash
set -u
TAG=${1:-r04m}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --timeout 180 > $O/${TAG}_pytest.txt 2>&1
tail -4 $O/${TAG}_pytest.txt
cd /tmp
for rep in 1 2; do
timeout 300 python $R/bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['secondary']; print(json.dumps({'c3':d['value'],'sec_ms':s['value'],'crop_ms':s['render_960x540_then_crop_ms'],'stages':{k:v['ms'] for k,v in s['stages_ms'].items()}}))" >> $O/${TAG}_secondary.jsonl
done
timeout 300 python $R/tools/k7_knockout.py 128 > $O/${TAG}_k7_knockouts.jsonl 2>/dev/null
timeout 300 python $R/tools/piece_probe.py --pieces 256,128,64 --no-c3 > $O/${TAG}_piece_probe.jsonl 2>/dev/null
echo done
