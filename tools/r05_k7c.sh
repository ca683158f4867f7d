#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-r05e}
: > $O/${TAG}_k7_knockouts.jsonl
for fb in 256 0 128 512 1024; do
  echo "{\"GCR_FILL_BLOCKS\": $fb}" >> $O/${TAG}_k7_knockouts.jsonl
  GCR_FILL_BLOCKS=$fb timeout 300 python tools/k7_knockout.py 128 >> $O/${TAG}_k7_knockouts.jsonl 2>/dev/null
done
cat $O/${TAG}_k7_knockouts.jsonl
