#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-r05c}
timeout 1100 python -m pytest tests -m gpu -q --timeout 400 -p no:cacheprovider -x > $O/${TAG}_pytest.txt 2>&1
tail -8 $O/${TAG}_pytest.txt
timeout 300 python tools/k7_knockout.py 128 > $O/${TAG}_k7_knockouts_item.jsonl 2>/dev/null; cat $O/${TAG}_k7_knockouts_item.jsonl
