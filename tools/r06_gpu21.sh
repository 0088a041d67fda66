#!/bin/bash
# round 6, GPU call 21: need buckets half an octave apart - 140 Mb (free, under 140 GB), one rank's 400 Mb share of configs[4] (free, under 160 GB)
set -u
O=gpurun_out/r06_21
mkdir -p $O
export HASLR_BENCH_DIR=/tmp/haslr_bench
AB_WORKLOAD=fly AB_PASSES=3 timeout 900 python tools/dev_r05_ab.py - poa_bucket_half_octaves=0 poa_workspace_gb=140 poa_workspace_gb=140,poa_bucket_half_octaves=0 - 2>&1 | grep RESULT | cut -c1-330 | tee $O/fly_ab.txt
rm -rf /tmp/haslr_bench
for v in "A=1" "HX_POA_BUCKET_HALF_OCTAVES=0" "HX_POA_WORKSPACE_GB=160" "HX_POA_WORKSPACE_GB=160 HX_POA_BUCKET_HALF_OCTAVES=0"; do
  echo "== $v"
  env $v timeout 900 python tools/full_size_check.py chm1_eighth --no-identity --no-sample --reuse 2>&1 | grep -E "gpu pass" | cut -c1-200
done | tee $O/eighth_ab.txt
rm -rf /tmp/full_size
