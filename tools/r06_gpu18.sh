#!/bin/bash
# round 6, GPU call 18: slots of the need buckets by work against top-down - 140 Mb (workspace left alone / capped) and one rank's 400 Mb share of configs[4]
set -u
O=gpurun_out/r06_18
mkdir -p $O
export HASLR_BENCH_DIR=/tmp/haslr_bench
AB_WORKLOAD=fly AB_PASSES=3 timeout 900 python tools/dev_r05_ab.py - poa_slots_by_work=0 poa_workspace_gb=140 poa_workspace_gb=140,poa_slots_by_work=0 poa_workspace_gb=100 poa_workspace_gb=70 2>&1 | grep RESULT | cut -c1-330 | tee $O/fly_slots_ab.txt
rm -rf /tmp/haslr_bench
for v in 1 0; do
  HX_POA_SLOTS_BY_WORK=$v HX_DEBUG=1 timeout 900 python tools/full_size_check.py chm1_eighth --no-identity --no-sample --reuse > $O/eighth_$v.json 2> $O/eighth_$v.err
  grep -E "gpu pass|POA batch" $O/eighth_$v.err | tail -4 | cut -c1-1500
done
HX_POA_WORKSPACE_GB=160 timeout 900 python tools/full_size_check.py chm1_eighth --no-identity --no-sample --reuse 2>&1 | grep -E "gpu pass" | cut -c1-300
rm -rf /tmp/full_size
