#!/bin/bash
# round 6, GPU call 24: the longest chains of the many-edge call with 4 instead of 8 columns per lane (twice the members)
set -u
O=gpurun_out/r06_24
mkdir -p $O
export HASLR_BENCH_DIR=/tmp/haslr_bench
AB_WORKLOAD=fly AB_PASSES=3 timeout 1200 python tools/dev_r05_ab.py - poa_cols2_top=8 poa_cols2_top=32 poa_cols2_top=32,poa_chain_ms=270 poa_cols2_top=32,poa_chain_ms=240 poa_cluster_max=12 poa_cluster_max=16 - 2>&1 | grep RESULT | cut -c1-330 | tee $O/fly_cols.txt
rm -rf /tmp/haslr_bench
