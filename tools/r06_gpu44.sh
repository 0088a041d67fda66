#!/bin/bash
# round 6, GPU call 44: 140 Mb with the own-bucket-first pick - larger chain caps, five passes each
set -u
O=gpurun_out/r06_44
mkdir -p $O
export HASLR_BENCH_DIR=/tmp/haslr_bench
AB_WORKLOAD=fly AB_PASSES=5 HX_DEBUG=1 timeout 1700 python tools/dev_r05_ab.py poa_chain_ms=440 poa_chain_ms=500 poa_chain_ms=560 poa_chain_ms=640 poa_chain_ms=800 poa_chain_ms=500 - 2>&1 | grep -E "RESULT|\] pass|column passes" | cut -c1-330 | uniq | tee $O/fly_sweep.txt
rm -rf /tmp/haslr_bench
