#!/bin/bash
# round 6, GPU call 8: default bench line (yeast main + configs1 + configs3 + one-shot figures), the group code at N = 1 and a host-staged --gpus 2 rehearsal
set -u
O=gpurun_out/r06_8
mkdir -p $O
export HASLR_BENCH_DIR=/tmp/haslr_bench
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
tail -2 $O/bench.err | cut -c1-300
HASLR_BENCH_FORCE_GROUP=1 HASLR_GROUP_TRANSPORT=host timeout 600 python bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_group1.json 2> $O/bench_group1.err
grep "step " $O/bench_group1.err | cut -c1-300
HASLR_GROUP_TRANSPORT=host timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_group2_host.json 2> $O/bench_group2_host.err
tail -3 $O/bench_group2_host.err | cut -c1-300
