#!/bin/bash
# round 6, GPU call 10: launch lists by chain rows (12 Mb / 4.6 Mb A/B in one process), the group code at N = 1 against the plain step, edge timeline
set -u
O=gpurun_out/r06_10
mkdir -p $O
export HASLR_BENCH_DIR=/tmp/haslr_bench
AB_WORKLOAD=yeast AB_PASSES=5 timeout 600 python tools/dev_r05_ab.py - poa_order_by_cells=1 - poa_order_by_cells=1 2>&1 | grep RESULT | cut -c1-250
AB_WORKLOAD=ecoli AB_PASSES=5 timeout 600 python tools/dev_r05_ab.py - poa_order_by_cells=1 2>&1 | grep RESULT | cut -c1-250
HASLR_BENCH_FORCE_GROUP=1 HASLR_GROUP_TRANSPORT=host timeout 600 python bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_group1.json 2> $O/bench_group1.err
grep "step " $O/bench_group1.err | cut -c1-100
