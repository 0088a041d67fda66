#!/bin/bash
# round 6, GPU call 43: 140 Mb with the own-bucket-first pick - the chain cap (the narrowest workgroup whose estimated chain stays below it), five passes each
set -u
O=gpurun_out/r06_43
mkdir -p $O
export HASLR_BENCH_DIR=/tmp/haslr_bench
AB_WORKLOAD=fly AB_PASSES=5 timeout 1700 python tools/dev_r05_ab.py - poa_chain_ms=320 poa_chain_ms=336 poa_chain_ms=350 poa_chain_ms=370 poa_chain_ms=400 poa_chain_ms=440 poa_chain_ms=336,poa_balance_pct=150 poa_chain_ms=336,poa_ring_kb=13 poa_chain_ms=336 - 2>&1 | grep -E "RESULT|pass" | cut -c1-330 | tee $O/fly_sweep.txt
rm -rf /tmp/haslr_bench
