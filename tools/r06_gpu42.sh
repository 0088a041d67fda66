#!/bin/bash
# round 6, GPU call 42: 140 Mb with the own-bucket-first pick - the launch-shape choices whose earlier A/Bs had the slow passes in them, five passes each
set -u
O=gpurun_out/r06_42
mkdir -p $O
export HASLR_BENCH_DIR=/tmp/haslr_bench
AB_WORKLOAD=fly AB_PASSES=5 timeout 1700 python tools/dev_r05_ab.py - poa_chain_ms=290 poa_chain_ms=336 poa_balance_pct=100 poa_balance_pct=150 poa_prune=92 poa_prune=97 poa_ring_kb=9 poa_ring_kb=13 poa_slots_pct=125 poa_slots_pct=80 poa_cols2_top=4 poa_wide_members=4 - 2>&1 | grep -E "RESULT" | cut -c1-330 | tee $O/fly_sweep.txt
rm -rf /tmp/haslr_bench
