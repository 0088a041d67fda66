#!/bin/bash
# development: A/B of run-time knobs on the bench data set; args = "ENV=VAL,ENV=VAL" sets ("-" = none)
cd $GRAFT_REPO_ROOT
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-configs1 > /dev/null 2>&1
for rep in 1 2; do
for a in "$@"; do
  e=""; [ "$a" != "-" ] && e=$(echo $a | tr ',' ' ')
  echo "== $e"
  env $e python bench.py --steps 4 --warmup 1 --no-cpu-baseline ${AB_ARGS:---no-configs1} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['ms_per_step'],1), d['poa_phase_cycles']['slowest_edge'], d['assembly']['sha256'][:12], (d.get('configs1') or {}).get('ms_per_step'))"
done
done
