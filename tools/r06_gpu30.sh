#!/bin/bash
# round 6, GPU call 30: 140 Mb, no cap: buckets an octave / half an octave apart, chain cap factor - alternating, five passes each
set -u
O=gpurun_out/r06_30
mkdir -p $O
export HASLR_BENCH_DIR=/tmp/haslr_bench
AB_WORKLOAD=fly AB_PASSES=5 timeout 1500 python tools/dev_r05_ab.py - poa_bucket_half_octaves=0 - poa_bucket_half_octaves=0 poa_chain_ms=336 poa_bucket_half_octaves=0,poa_chain_ms=336 - 2>&1 | grep -E "RESULT|pass" | cut -c1-330 | tee $O/fly_ab.txt
rm -rf /tmp/haslr_bench
