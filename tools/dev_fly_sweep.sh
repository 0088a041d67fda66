#!/bin/bash
# throughput-regime sweep on the 140 Mb data set: columns per lane x ring budget per wave
cd $GRAFT_REPO_ROOT
python tools/full_size_check.py fly --no-oracle --no-identity --passes 1 > /dev/null 2>&1
for cfg in "$@"; do
  set -- $(echo $cfg | tr ':' ' ')
  echo "== COLS=$1 RING_KB=$2 WAVE_MAX=${3:-512}"
  HX_DEBUG=1 HX_POA_COLS=$1 HX_POA_RING_KB=$2 HX_POA_WAVE_MAX=${3:-512} python tools/full_size_check.py fly --no-oracle --no-identity --reuse 2>&1 | grep -E "gpu pass 1|POA batch: [0-9]+ edges|class [0-9]+|all edges|top edge" | tail -24
done
