#!/bin/bash
set -u
TAG=${1:-r04j}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --timeout 180 > $O/${TAG}_pytest.txt 2>&1
tail -4 $O/${TAG}_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.txt 2>&1
tail -3 $O/${TAG}_smoke.txt
cd /tmp; timeout 300 python $R/bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_driver_like.json 2>/dev/null
echo done
