#!/bin/bash
# round 6, GPU call 41: one rank's 400 Mb share of configs[4] - persistent workgroups taking their own need bucket first / the longest chain of any bucket
set -u
O=gpurun_out/r06_41
mkdir -p $O
for v in "A=1" "HX_POA_OWN_BUCKET_FIRST=0" "A=2"; do
  echo "== $v"
  env $v timeout 900 python tools/full_size_check.py chm1_eighth --no-identity --no-sample --reuse --passes 4 2>&1 | grep -E "gpu pass" | cut -c1-200
done | tee $O/eighth_ab.txt
rm -rf /tmp/full_size
