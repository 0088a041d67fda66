#!/bin/bash
# round 6, GPU call 20: the chain cap of the many-edge regime again (140 Mb), shared-edge count / members
set -u
O=gpurun_out/r06_20
mkdir -p $O
export HASLR_BENCH_DIR=/tmp/haslr_bench
AB_WORKLOAD=fly AB_PASSES=3 timeout 1200 python tools/dev_r05_ab.py - poa_chain_ms=260 poa_chain_ms=290 poa_chain_ms=315 poa_chain_ms=340 poa_chain_ms=370 poa_cluster_topk=48 poa_cluster_topk=16 poa_cluster_max=12 - 2>&1 | grep RESULT | cut -c1-330 | tee $O/fly_cap.txt
rm -rf /tmp/haslr_bench
