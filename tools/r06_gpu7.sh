#!/bin/bash
# round 6, GPU call 7: 2 columns per lane for the members of the K costliest shared edges (12 Mb / 4.6 Mb), then the GPU suite without the 400 Mb test
set -u
O=gpurun_out/r06_7
mkdir -p $O
export HASLR_BENCH_DIR=/dev/shm/haslr_bench
AB_WORKLOAD=yeast AB_PASSES=4 timeout 900 python tools/dev_r05_ab.py - poa_cols2_top=1 poa_cols2_top=2 poa_cols2_top=4 poa_cols2_top=8 poa_cols2_top=16 poa_cols2_top=32 - 2>&1 | grep -E "RESULT" > $O/cols2.txt
AB_WORKLOAD=ecoli AB_PASSES=4 timeout 900 python tools/dev_r05_ab.py - poa_cols2_top=2 poa_cols2_top=4 poa_cols2_top=8 2>&1 | grep -E "RESULT" >> $O/cols2.txt
cat $O/cols2.txt | cut -c1-300
rm -rf /dev/shm/haslr_bench
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 -k "not configs4" > $O/gpu_tests_main.log 2>&1
tail -25 $O/gpu_tests_main.log
df -h / > $O/df_after.txt
