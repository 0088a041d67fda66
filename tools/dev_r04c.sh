#!/bin/bash
# development: A/B of builds (arguments as tools/dev_ab.sh) + the consensus tests and a fuzz session of the in-tree build, in one GPU call
cd $GRAFT_REPO_ROOT
T=${TAG:-r04c}
mkdir -p gpurun_out/$T
AB_ARGS=" " bash tools/dev_ab.sh "$@" 2>&1 | tee gpurun_out/$T/ab.txt
HASLR_SKIP_HUGE=1 timeout 900 python -m pytest tests -m gpu -x -q -k "poa or stage or golden or spoa or cluster or wide or block" > gpurun_out/$T/pytest.log 2>&1; tail -n 3 gpurun_out/$T/pytest.log
timeout 600 python tools/dev_fuzz.py ${FUZZ_N:-40} ${FUZZ_SEED:-7072} > gpurun_out/$T/fuzz.txt 2>&1; tail -n 1 gpurun_out/$T/fuzz.txt; grep -c " OK " gpurun_out/$T/fuzz.txt
