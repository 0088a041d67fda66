#!/bin/bash
# development: a longer fuzz session (big and small cases, fresh seeds)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
FUZZ_BIG=1 timeout 2400 python tools/dev_fuzz.py ${1:-40} ${2:-5051} > gpurun_out/fuzz_big2.txt 2>&1; tail -n 1 gpurun_out/fuzz_big2.txt; grep -c " OK " gpurun_out/fuzz_big2.txt
timeout 2400 python tools/dev_fuzz.py ${3:-150} ${4:-5052} > gpurun_out/fuzz_small2.txt 2>&1; tail -n 1 gpurun_out/fuzz_small2.txt; grep -c " OK " gpurun_out/fuzz_small2.txt
grep -v " OK " gpurun_out/fuzz_big2.txt gpurun_out/fuzz_small2.txt | head -20
