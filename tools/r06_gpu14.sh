#!/bin/bash
# round 6, GPU call 14: exact pruning in the shared (cluster) edges - the tests that force cluster shapes, then the 12 Mb / 4.6 Mb calls with and without it
set -u
O=gpurun_out/r06_14
mkdir -p $O
cat /sys/fs/cgroup/memory.max /sys/fs/cgroup/memory.current > $O/cgroup_memory.txt 2>&1
export HASLR_BENCH_DIR=/tmp/haslr_bench
HASLR_SKIP_HUGE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cluster or far_row or random or stall or gap_longer or every_stage or persistent or need_buckets or in_degree or known or one_workgroup or seeds or score_matrix" > $O/tests_cluster.log 2>&1
tail -5 $O/tests_cluster.log
AB_WORKLOAD=yeast AB_PASSES=4 timeout 600 python tools/dev_r05_ab.py - poa_prune_shared=0 - poa_prune_shared=0 2>&1 | grep RESULT | cut -c1-330
AB_WORKLOAD=ecoli AB_PASSES=4 timeout 600 python tools/dev_r05_ab.py - poa_prune_shared=0 2>&1 | grep RESULT | cut -c1-330
