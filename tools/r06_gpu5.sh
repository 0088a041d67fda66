#!/bin/bash
# round 6, GPU call 5: the row experiments on the 12 Mb data set (the lone wave's chain): fast-row loop against the one-loop build, 2 columns per lane for the
# members of shared edges; then the GPU suite without the 400 Mb test
set -u
O=gpurun_out/r06_5
mkdir -p $O
export HASLR_BENCH_DIR=/dev/shm/haslr_bench
for rep in 1 2; do
  AB_WORKLOAD=yeast AB_PASSES=5 timeout 600 python tools/dev_r05_ab.py - poa_cluster_cols=2 2>&1 | grep -E "RESULT|pass 4"
  AB_LIBDIR=haslr_amd/lib_nofast AB_WORKLOAD=yeast AB_PASSES=5 timeout 600 python tools/dev_r05_ab.py - poa_cluster_cols=2 2>&1 | grep -E "RESULT|pass 4" | sed 's/^/nofast /'
done > $O/row_ab.txt 2>&1
AB_WORKLOAD=ecoli AB_PASSES=4 timeout 600 python tools/dev_r05_ab.py - poa_cluster_cols=2 2>&1 | grep -E "RESULT" >> $O/row_ab.txt
AB_LIBDIR=haslr_amd/lib_nofast AB_WORKLOAD=ecoli AB_PASSES=4 timeout 600 python tools/dev_r05_ab.py - 2>&1 | grep -E "RESULT" | sed 's/^/nofast /' >> $O/row_ab.txt
cat $O/row_ab.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 --basetemp=/dev/shm/pt -k "not configs4" > $O/gpu_tests_main.log 2>&1
tail -4 $O/gpu_tests_main.log
rm -rf /dev/shm/pt /dev/shm/haslr_bench
