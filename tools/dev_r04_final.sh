#!/bin/bash
# development: the round's closing GPU call - whole GPU suite, fuzz sessions, the profile set of both workloads (tools/dev_profile_run.sh)
cd "$GRAFT_REPO_ROOT"
T=${1:-r04_v3}
mkdir -p gpurun_out/$T
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 > gpurun_out/$T/gpu_tests_all.log 2>&1; tail -n 3 gpurun_out/$T/gpu_tests_all.log
timeout 900 python tools/dev_fuzz.py ${2:-120} ${3:-8081} > gpurun_out/$T/fuzz.txt 2>&1; tail -n 1 gpurun_out/$T/fuzz.txt; grep -c " OK " gpurun_out/$T/fuzz.txt
FUZZ_BIG=1 timeout 900 python tools/dev_fuzz.py ${4:-30} ${5:-8082} > gpurun_out/$T/fuzz_big.txt 2>&1; tail -n 1 gpurun_out/$T/fuzz_big.txt; grep -c " OK " gpurun_out/$T/fuzz_big.txt
bash tools/dev_profile_run.sh $T prof 2>&1 | tail -25
bash tools/dev_profile_run.sh $T fly 2>&1 | tail -25
