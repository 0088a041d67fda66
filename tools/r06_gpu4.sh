#!/bin/bash
# round 6, GPU call 4: the GPU suite (durations), what is given as $1 is passed to pytest -k
set -u
O=gpurun_out/r06_4
mkdir -p $O
df -h / /dev/shm > $O/df_before.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 --basetemp=/dev/shm/pt -k "$1" > $O/gpu_tests_$2.log 2>&1
tail -4 $O/gpu_tests_$2.log
df -h / /dev/shm > $O/df_after.txt 2>&1
rm -rf /dev/shm/pt
