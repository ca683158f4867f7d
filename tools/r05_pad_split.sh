#!/bin/bash
# K6 capped at 7 workgroups per CU by ONE extra KB of LDS (eight workgroups take every wave slot of a CU: nothing of another
# frame can be resident beside them) x K1 as two kernels (the streaming cull has 55 VGPRs and 264 B of LDS: it fits the
# one wave slot per SIMD that seven blend workgroups leave).  Same box, alternating.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-r05n}
: > $O/${TAG}_k6_cap7_split_k1.jsonl
B="python bench.py --no-cpu-baseline --no-secondary"
for rep in 1 2; do
for v in "0:" "0:--split-preprocess" "1024:" "1024:--split-preprocess" "2048:--split-preprocess" "1024:--split-preprocess --streams 2" "1024:--split-preprocess --streams 4"; do
  pad=${v%%:*}; fl=${v#*:}
  echo "{\"GCR_K6_LDS_PAD\": $pad, \"flags\": \"$fl\"}" >> $O/${TAG}_k6_cap7_split_k1.jsonl
  GCR_LIB_PATH=$R/tools/_build/libgcr_hip_exp.so GCR_K6_LDS_PAD=$pad timeout 200 $B $fl 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps({'value':d['value'],'other':d.get('other_entry_point',{}).get('value'),'stages':{k:[v['ms'],v['ms_single_stream']] for k,v in d['stages_ms'].items()}}))" >> $O/${TAG}_k6_cap7_split_k1.jsonl
done; done
cat $O/${TAG}_k6_cap7_split_k1.jsonl
