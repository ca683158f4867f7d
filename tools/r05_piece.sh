#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-r05g}
: > $O/${TAG}_infer_piece.jsonl
for cfg in C5 C3 D1; do
for pc in 223 160 128 96 223; do
  echo "{\"config\": \"$cfg\", \"GCR_INFER_PIECE\": $pc}" >> $O/${TAG}_infer_piece.jsonl
  GCR_LIB_PATH=$R/tools/_build/libgcr_hip_exp.so GCR_INFER_PIECE=$pc timeout 200 python bench.py --config $cfg --steps 48 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps({'value':d['value'],'ms':d['ms_per_step'],'stages':{k:[v['ms'],v['ms_single_stream']] for k,v in d['stages_ms'].items()}}))" >> $O/${TAG}_infer_piece.jsonl
done; done
cat $O/${TAG}_infer_piece.jsonl
