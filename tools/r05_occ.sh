#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-r05h}
: > $O/${TAG}_k7_occ_ab.jsonl
for rep in 1 2; do
for v in ship occ6; do
  if [ $v = ship ]; then env -u GCR_LIB_PATH timeout 300 python tools/k7_ab.py C2 "0:128,0:96,0:160" >> $O/${TAG}_k7_occ_ab.jsonl 2>/dev/null
  else GCR_LIB_PATH=$R/tools/_build/libgcr_hip_$v.so timeout 300 python tools/k7_ab.py C2 "0:128,0:96,0:160" >> $O/${TAG}_k7_occ_ab.jsonl 2>/dev/null; fi
done; done
cat $O/${TAG}_k7_occ_ab.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['lib'], d['bwd_piece'], d['blend_fwd_ms'], d['blend_bwd_ms'], d['fwd_bwd_wall_ms'], '%.1e'%d['worst_rel_diff_vs_first'])"
