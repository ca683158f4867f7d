#!/bin/bash
# A/B of two builds of libgcr_hip.so on one box: tools/ab_libs.sh OUTDIR "bench args" name=path [name=path ...]
# (GCR_LIB_PATH selects the library the package loads; one JSON line per run under OUTDIR).
out=$1; shift; bargs=$1; shift
mkdir -p "$out"
for rep in 1 2; do
  for kv in "$@"; do
    name=${kv%%=*}; path=${kv#*=}
    tag=$(echo "$bargs" | tr -c 'A-Za-z0-9\n' '_')
    GCR_LIB_PATH=$path python bench.py $bargs --no-cpu-baseline --no-secondary >> "$out/${name}${tag}.jsonl" 2>> "$out/${name}${tag}.err"
  done
done
