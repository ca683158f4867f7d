#!/bin/bash
# A/B helper: every line of the file given as $1 is a set of bench.py arguments; prints value / period / latency / stages
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
TAG=${2:-r06_ab}
B="python bench.py --no-cpu-baseline --no-secondary"
while IFS= read -r o; do
  [ -z "$o" ] && continue
  $B $o 2>$O/${TAG}_err.txt | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('''$o''', '|', d['value'], d['ms_per_step'], d['frame_latency_ms'], d.get('other_entry_point',{}).get('value'), {k:(v['ms'],v['ms_single_stream']) for k,v in d['stages_ms'].items() if k!='ranges'})
except Exception as e:
    print('''$o''', 'FAILED', e); print(open('$O/${TAG}_err.txt').read()[-1500:])"
done < $1 > $O/$TAG.txt 2>&1
cat $O/$TAG.txt
