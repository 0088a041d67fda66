#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-fly_trace}
mkdir -p $O
python $R/tools/full_size_check.py fly --no-oracle --no-identity --passes 1 > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/tools/full_size_check.py fly --no-oracle --no-identity --reuse > $O/kt.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) > $O/kernel_stats.txt 2>&1
grep -A40 "k_poa dispatches" $O/kernel_stats.txt | tail -45
grep "gpu pass" $O/kt.log
