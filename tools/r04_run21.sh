#!/bin/bash
set -u
TAG=${1:-r04y}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_dma_probe tools/lds_dma_probe.hip 2>/dev/null && timeout 60 /tmp/lds_dma_probe > $O/${TAG}_lds_dma_probe.txt 2>&1
cat $O/${TAG}_lds_dma_probe.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 180 -x > $O/${TAG}_pytest.txt 2>&1
tail -5 $O/${TAG}_pytest.txt
: > $O/${TAG}_ab.jsonl
for rep in 1 2; do
  for v in ship occ7; do
    echo "{\"variant\": \"K6 $v\"}" >> $O/${TAG}_ab.jsonl
    if [ $v = ship ]; then timeout 300 python bench.py --steps 400 --warmup 50 2>/dev/null >> $O/${TAG}_ab.jsonl
    else GCR_LIB_PATH=$R/tools/_build/libgcr_hip_$v.so timeout 300 python bench.py --steps 400 --warmup 50 2>/dev/null >> $O/${TAG}_ab.jsonl; fi
  done
done
for v in ship occ7; do
  echo "{\"variant\": \"K6 $v\"}" >> $O/${TAG}_ab.jsonl
  if [ $v = ship ]; then timeout 300 python tools/piece_probe.py --pieces 128 2>/dev/null >> $O/${TAG}_ab.jsonl
  else GCR_LIB_PATH=$R/tools/_build/libgcr_hip_$v.so timeout 300 python tools/piece_probe.py --pieces 128 2>/dev/null >> $O/${TAG}_ab.jsonl; fi
done
timeout 300 python tools/k6_clocks.py > $O/${TAG}_k6_clocks.jsonl 2>/dev/null
cat $O/${TAG}_k6_clocks.jsonl
echo done
