#!/bin/bash
# latency-regime sweep on the bench data set: cluster member shape
cd $GRAFT_REPO_ROOT
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-configs1 > /dev/null 2>&1
for cfg in "$@"; do
  set -- $(echo $cfg | tr ':' ' ')
  echo "== MEMBER_LANES=$1 CLUSTER_MAX=$2 CLUSTER_COLS=${3:-4} TOPK=${4:-96} CLUSTER_MIN=${5:-2048}"
  HX_POA_MEMBER_LANES=$1 HX_POA_CLUSTER_MAX=$2 HX_POA_CLUSTER_COLS=${3:-4} HX_POA_CLUSTER_TOPK=${4:-96} HX_POA_CLUSTER_MIN=${5:-2048} python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['ms_per_step'],1), d['poa_phase_cycles']['slowest_edge'])"
done
