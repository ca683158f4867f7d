#!/bin/bash
# experiment build: blends of consecutive frames chained by events (GCR_CHAIN_BLEND) x the blend at 7 workgroups per CU
# (GCR_K6_LDS_PAD=2048) x fused / split K1 x stream counts
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-r05o}
: > $O/${TAG}_blend_chain.jsonl
B="python bench.py --no-cpu-baseline --no-secondary"
for v in "0:0:" "1:0:" "1:2048:" "1:2048:--split-preprocess" "1:0:--split-preprocess" "1:2048:--split-preprocess --streams 2" "1:2048:--streams 2" "1:2048:--split-preprocess --streams 4" "0:0:"; do
  ch=${v%%:*}; rest=${v#*:}; pad=${rest%%:*}; fl=${rest#*:}
  echo "{\"GCR_CHAIN_BLEND\": $ch, \"GCR_K6_LDS_PAD\": $pad, \"flags\": \"$fl\"}" >> $O/${TAG}_blend_chain.jsonl
  GCR_LIB_PATH=$R/tools/_build/libgcr_hip_exp.so GCR_CHAIN_BLEND=$ch GCR_K6_LDS_PAD=$pad timeout 200 $B $fl 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps({'value':d['value'],'other':d.get('other_entry_point',{}).get('value'),'stages':{k:[v['ms'],v['ms_single_stream']] for k,v in d['stages_ms'].items()}}))" >> $O/${TAG}_blend_chain.jsonl
done
cat $O/${TAG}_blend_chain.jsonl
