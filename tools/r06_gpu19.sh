#!/bin/bash
# round 6, GPU call 19: slot policy by budget (140 Mb free / capped), far-row estimate of the rings of 4 (400 Mb share under 160 GB)
set -u
O=gpurun_out/r06_19
mkdir -p $O
export HASLR_BENCH_DIR=/tmp/haslr_bench
AB_WORKLOAD=fly AB_PASSES=3 timeout 900 python tools/dev_r05_ab.py - poa_far_shift=4 poa_far_shift=5 poa_workspace_gb=140 poa_workspace_gb=140,poa_far_shift=4 poa_workspace_gb=140,poa_far_shift=5 2>&1 | grep RESULT | cut -c1-330 | tee $O/fly_ab.txt
rm -rf /tmp/haslr_bench
for v in "HX_POA_WORKSPACE_GB=160" "HX_POA_WORKSPACE_GB=160 HX_POA_FAR_SHIFT=4" "HX_POA_WORKSPACE_GB=160 HX_POA_FAR_SHIFT=5" "HX_POA_FAR_SHIFT=4"; do
  echo "== $v"
  env $v HX_DEBUG=1 timeout 900 python tools/full_size_check.py chm1_eighth --no-identity --no-sample --reuse 2>&1 | grep -E "gpu pass|to be redone" | cut -c1-260
done | tee $O/eighth_ab.txt
rm -rf /tmp/full_size
