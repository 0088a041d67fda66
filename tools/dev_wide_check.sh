#!/bin/bash
cd $GRAFT_REPO_ROOT
for g in 4600000 12000000 20000000 30000000; do
  for w in "" 0 4; do
    if [ -z "$w" ]; then unset HX_POA_WIDE_MEMBERS; else export HX_POA_WIDE_MEMBERS=$w; fi
    m=pacbio; 
    r=$(HX_DEBUG=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --genome-len $g 2> /tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['ms_per_step'],1), d['config']['edges'])")
    echo "genome $g wide=${w:-auto}: $r $(grep 'wide members' /tmp/err.txt | tail -n 1)"
  done
done
