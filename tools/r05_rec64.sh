#!/bin/bash
# 64-byte records (ship) against the 48-byte layout (tools/_build/libgcr_hip_rec48.so): tests, K1 / frame A/B, counted bytes
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 400 -x 2>&1 | tail -3 > $O/r05_rec64_pytest.txt
cat $O/r05_rec64_pytest.txt
grep -q " passed" $O/r05_rec64_pytest.txt || exit 1
grep -q "failed" $O/r05_rec64_pytest.txt && exit 1
REPS=3 bash tools/ab_variants.sh r05_rec64 "python tools/k1_ab.py C3" ship rec48 > /dev/null
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-secondary --steps 8 --warmup 2"
for v in ship rec48; do
  if [ $v = ship ]; then unset GCR_LIB_PATH; else export GCR_LIB_PATH=$R/tools/_build/libgcr_hip_$v.so; fi
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/rec_${v}_f -o p -- $B > /dev/null 2>&1 < /dev/null
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/rec_${v}_w -o p -- $B > /dev/null 2>&1 < /dev/null
  python $R/tools/make_traffic.py /tmp/rec_${v}_f/p_results.db /tmp/rec_${v}_w/p_results.db $O/r05_traffic_records_${v}.json "" "C3 forward, record layout A/B: $v" > /dev/null
done
unset GCR_LIB_PATH
echo done
