#!/bin/bash
# round 6, GPU call 45: the automatic chain cap as a percentage of the estimated wave-slot time (poa_chain_pct) - 140 Mb and one rank's 400 Mb share of configs[4]
set -u
O=gpurun_out/r06_45
mkdir -p $O
export HASLR_BENCH_DIR=/tmp/haslr_bench
AB_WORKLOAD=fly AB_PASSES=1 HX_DEBUG=2 timeout 300 python tools/dev_r05_ab.py - 2>&1 | grep -E "chain cap" | sort -u | sort -t' ' -k4 -n | tee $O/fly_caps.txt
AB_WORKLOAD=fly AB_PASSES=5 timeout 1700 python tools/dev_r05_ab.py - poa_chain_pct=70 poa_chain_pct=80 poa_chain_pct=90 poa_chain_pct=100 poa_chain_pct=80 - 2>&1 | grep -E "RESULT|\] pass" | cut -c1-330 | tee $O/fly_sweep.txt
rm -rf /tmp/haslr_bench
for v in "HX_POA_CHAIN_PCT=60" "HX_POA_CHAIN_PCT=80" "HX_POA_CHAIN_PCT=100"; do
  echo "== $v"
  env $v HX_DEBUG=1 timeout 900 python tools/full_size_check.py chm1_eighth --no-identity --no-sample --reuse --passes 3 2>&1 | grep -E "gpu pass|column passes" | cut -c1-200 | uniq
done | tee $O/eighth_ab.txt
rm -rf /tmp/full_size
