#!/bin/bash
# round 5: full GPU test suite, then the randomised parity sweep over fresh seeds with the round-5 kernels
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-r05i}
timeout 1100 python -m pytest tests -m gpu -q --timeout 400 -p no:cacheprovider > $O/${TAG}_pytest.txt 2>&1
tail -6 $O/${TAG}_pytest.txt
: > $O/${TAG}_fuzz.jsonl
for seed in 51 52 53 54; do
  timeout 260 python tools/fuzz_parity.py --seed $seed --cases 1500 --max-seconds 110 --wrapper-cases 100 >> $O/${TAG}_fuzz.jsonl 2>/dev/null
done
grep -c . $O/${TAG}_fuzz.jsonl; grep '"cases"\|wrapper_cases' $O/${TAG}_fuzz.jsonl; grep '"case"' $O/${TAG}_fuzz.jsonl | cut -c1-300
