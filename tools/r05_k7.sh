#!/bin/bash
# round 5: GPU tests + the K7 A/B (two kernels x piece sizes) + the float64 attribution of the recorded fuzz exceedances
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-r05b}
timeout 1100 python -m pytest tests -m gpu -q --timeout 400 -p no:cacheprovider > $O/${TAG}_pytest.txt 2>&1
tail -25 $O/${TAG}_pytest.txt
timeout 300 python tools/k7_ab.py C2 "${2:-1:128,0:128,0:96,0:160,0:223,1:223,1:128,0:128}" > $O/${TAG}_k7_ab.jsonl 2> $O/${TAG}_k7_ab.err
cat $O/${TAG}_k7_ab.jsonl; tail -5 $O/${TAG}_k7_ab.err
GCR_LIB_PATH=$R/tools/_build/libgcr_hip_exp.so timeout 300 python tools/fuzz_f64.py 48:469 46:368 43:95 > $O/${TAG}_fuzz_f64.jsonl 2> $O/${TAG}_fuzz_f64.err
tail -3 $O/${TAG}_fuzz_f64.err; wc -l $O/${TAG}_fuzz_f64.jsonl
