#!/bin/bash
# blends chained across frames x the blend at 4 / 5 / 6 waves per SIMD by register padding (LDS stays free for the other
# frames' kernels, whose workgroups then fit beside a blend)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-r05p}
: > $O/${TAG}_blend_chain_waves.jsonl
B="python bench.py --no-cpu-baseline --no-secondary"
for v in "exp:0:" "k6w6:1:" "k6w5:1:" "k6w4:1:" "k6w6:0:" "k6w5:0:" "k6w4:0:" "k6w5:1:--streams 4" "k6w4:1:--streams 4" "k6w4:1:--streams 2" "exp:0:"; do
  lib=${v%%:*}; rest=${v#*:}; ch=${rest%%:*}; fl=${rest#*:}
  echo "{\"lib\": \"$lib\", \"GCR_CHAIN_BLEND\": $ch, \"flags\": \"$fl\"}" >> $O/${TAG}_blend_chain_waves.jsonl
  GCR_LIB_PATH=$R/tools/_build/libgcr_hip_$lib.so GCR_CHAIN_BLEND=$ch timeout 200 $B $fl 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps({'value':d['value'],'other':d.get('other_entry_point',{}).get('value'),'stages':{k:[v['ms'],v['ms_single_stream']] for k,v in d['stages_ms'].items()}}))" >> $O/${TAG}_blend_chain_waves.jsonl
done
cat $O/${TAG}_blend_chain_waves.jsonl
