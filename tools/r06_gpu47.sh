#!/bin/bash
# round 6, GPU call 47: 140 Mb with the new chain cap (70 %) - the other launch-shape choices once more, five passes each
set -u
O=gpurun_out/r06_47
mkdir -p $O
export HASLR_BENCH_DIR=/tmp/haslr_bench
AB_WORKLOAD=fly AB_PASSES=5 timeout 1700 python tools/dev_r05_ab.py - poa_balance_pct=100 poa_balance_pct=150 poa_prune=93 poa_prune=97 poa_ring_kb=9 poa_ring_kb=13 poa_slots_pct=125 poa_slots_pct=80 poa_workspace_gb=180 poa_workspace_gb=140 poa_cluster_topk=24 poa_cluster_topk=48 - 2>&1 | grep -E "RESULT" | cut -c1-330 | tee $O/fly_sweep.txt
rm -rf /tmp/haslr_bench
