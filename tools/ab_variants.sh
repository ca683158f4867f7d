#!/bin/bash
# A/B of the shipped library against variant builds on ONE box, alternating (gpurun boxes differ by a few per cent, so
# every comparison in profiles/r04_*_ab.jsonl / *_experiment.jsonl was taken this way).
#   make -C gaussiancity_amd/csrc BUILD=_build_x OUT=../../tools/_build/libgcr_hip_x.so EXTRA=-DSOME_MACRO=1 ../../tools/_build/libgcr_hip_x.so
#   gpurun -- 'bash tools/ab_variants.sh <tag> "<command>" ship x [y ...]'      e.g. command = "python bench.py --steps 400 --warmup 50"
# writes gpurun_out/<tag>_ab.jsonl: a {"variant": ...} line, then the command's output, three rounds.
set -u
TAG=$1; CMD=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out; mkdir -p $O
: > $O/${TAG}_ab.jsonl
for rep in $(seq 1 ${REPS:-3}); do
  for v in "$@"; do
    echo "{\"variant\": \"$v\"}" >> $O/${TAG}_ab.jsonl
    if [ $v = ship ]; then env -u GCR_LIB_PATH timeout 600 $CMD 2>/dev/null >> $O/${TAG}_ab.jsonl
    else GCR_LIB_PATH=$R/tools/_build/libgcr_hip_$v.so timeout 600 $CMD 2>/dev/null >> $O/${TAG}_ab.jsonl; fi
  done
done
echo done
