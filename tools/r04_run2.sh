#!/bin/bash
set -u
TAG=${1:-r04d}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --timeout 180 > $O/${TAG}_pytest.txt 2>&1
tail -4 $O/${TAG}_pytest.txt
for m in 0 1; do
  GCR_EXT_SYNC=$m timeout 200 python bench.py --inference-loop --steps 240 --host-camera closed-form > $O/${TAG}_inference_loop_closed_form_sync$m.json 2>/dev/null
  GCR_EXT_SYNC=$m timeout 200 python bench.py --train-step --steps 200 --host-camera closed-form > $O/${TAG}_c4_trainstep_closed_form_sync$m.json 2>/dev/null
done
bash tools/r04_sweep.sh $TAG
