#!/bin/bash
# round 6, GPU call 46: the automatic chain cap as a percentage of the estimated wave-slot time (poa_chain_pct) - one rank's 400 Mb share of configs[4], lower percentages
set -u
O=gpurun_out/r06_46
mkdir -p $O
for v in "HX_POA_CHAIN_PCT=40" "HX_POA_CHAIN_PCT=50" "HX_POA_CHAIN_PCT=70" "HX_POA_CHAIN_PCT=60"; do
  echo "== $v"
  env $v HX_DEBUG=1 timeout 900 python tools/full_size_check.py chm1_eighth --no-identity --no-sample --reuse --passes 3 2>&1 | grep -E "gpu pass|column passes|POA batch" | cut -c1-260 | uniq
done | tee $O/eighth_ab.txt
rm -rf /tmp/full_size
