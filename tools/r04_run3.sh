#!/bin/bash
set -u
TAG=${1:-r04e}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_async.py tests/test_gpu_fuzz.py -m gpu -q --timeout 180 > $O/${TAG}_pytest.txt 2>&1
tail -4 $O/${TAG}_pytest.txt
timeout 300 python tools/c4_leg_ab.py > $O/${TAG}_c4_leg_ab_closed_form.json 2>$O/${TAG}_err.txt
timeout 300 python tools/c4_leg_ab.py --camera reference > $O/${TAG}_c4_leg_ab_reference.json 2>>$O/${TAG}_err.txt
GCR_NO_AFFINITY=1 timeout 300 python bench.py --train-step --steps 200 --host-camera closed-form > $O/${TAG}_c4_trainstep_noaff.json 2>>$O/${TAG}_err.txt
timeout 300 python bench.py --train-step --steps 200 --host-camera closed-form > $O/${TAG}_c4_trainstep_aff.json 2>>$O/${TAG}_err.txt
echo done
