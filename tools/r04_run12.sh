#!/bin/bash
set -u
TAG=${1:-r04n}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
: > $O/${TAG}_k7_skip_ab.jsonl
for rep in 1 2 3; do
  echo '{"variant": "no wave-skip branch (shipping)"}' >> $O/${TAG}_k7_skip_ab.jsonl
  timeout 300 python tools/piece_probe.py --pieces 128 --no-c3 2>/dev/null >> $O/${TAG}_k7_skip_ab.jsonl
  echo '{"variant": "with the wave-skip branch"}' >> $O/${TAG}_k7_skip_ab.jsonl
  GCR_LIB_PATH=$R/tools/_build/libgcr_hip_skip.so timeout 300 python tools/piece_probe.py --pieces 128 --no-c3 2>/dev/null >> $O/${TAG}_k7_skip_ab.jsonl
done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q --timeout 180 -k "backward or c2 or pieces or c4 or fuzz or deterministic" 2>&1 | tail -3
echo done
