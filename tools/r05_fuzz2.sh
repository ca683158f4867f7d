#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-r05q}
: > $O/${TAG}_fuzz.jsonl
for seed in 55 56 57; do
  timeout 260 python tools/fuzz_parity.py --seed $seed --cases 1500 --max-seconds 110 --wrapper-cases 100 >> $O/${TAG}_fuzz.jsonl 2>/dev/null
done
grep '"cases"\|wrapper_cases' $O/${TAG}_fuzz.jsonl; grep '"case"' $O/${TAG}_fuzz.jsonl | cut -c1-400
