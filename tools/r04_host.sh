#!/bin/bash
# host-side profiles of the product's own loops (C4 leg, inference loop), with and without the host wait
set -u
TAG=${1:-r04c}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --timeout 180 > $O/${TAG}_pytest.txt 2>&1
tail -5 $O/${TAG}_pytest.txt
for m in 0 1; do
  GCR_EXT_SYNC=$m timeout 200 python tools/host_profile_c4.py --host-camera > $O/${TAG}_c4_host_profile_closed_form_sync$m.txt 2>&1
  GCR_EXT_SYNC=$m timeout 200 python tools/host_profile_c4.py > $O/${TAG}_c4_host_profile_reference_sync$m.txt 2>&1
  GCR_EXT_SYNC=$m timeout 200 python bench.py --inference-loop --steps 240 > $O/${TAG}_inference_loop_sync$m.json 2>/dev/null
  GCR_EXT_SYNC=$m timeout 200 python bench.py --inference-loop --steps 240 --host-camera closed-form > $O/${TAG}_inference_loop_closed_form_sync$m.json 2>/dev/null
  GCR_EXT_SYNC=$m timeout 200 python bench.py --train-step --steps 200 > $O/${TAG}_c4_trainstep_sync$m.json 2>/dev/null
done
echo done
