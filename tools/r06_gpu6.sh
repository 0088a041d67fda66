#!/bin/bash
# round 6, GPU call 6: where the rows of the two builds spend their cycles (-DHX_DP_PROF: segments of wave 0's rows per launch class)
set -u
O=gpurun_out/r06_6
mkdir -p $O
export HASLR_BENCH_DIR=/dev/shm/haslr_bench
for v in prof_fast prof_nofast; do
  AB_LIBDIR=haslr_amd/lib_$v AB_WORKLOAD=yeast AB_PASSES=2 HX_DEBUG=1 HX_PROF1=1 timeout 600 python tools/dev_r05_ab.py - 2>&1 | grep -E "RESULT|prof1|pass 1" | sed "s/^/$v /"
done > $O/prof.txt 2>&1
for v in lib lib_nofast; do
  AB_LIBDIR=haslr_amd/$v AB_WORKLOAD=yeast AB_PASSES=3 HX_DEBUG=1 timeout 600 python tools/dev_r05_ab.py - 2>&1 | grep -E "RESULT|top edge|slowest edge|class [0-9]*:" | tail -24 | sed "s/^/$v /"
done > $O/phases.txt 2>&1
cat $O/prof.txt $O/phases.txt | cut -c1-400
rm -rf /dev/shm/haslr_bench
