cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
AB_ARGS=" " bash tools/dev_ab.sh haslr_amd/lib_base - 2>&1 | tee gpurun_out/r04c/ab_deramp.txt
HASLR_SKIP_HUGE=1 timeout 900 python -m pytest tests -m gpu -x -q -k "poa or stage or golden or spoa or cluster or wide or block" > gpurun_out/r04c/pytest.log 2>&1; tail -n 3 gpurun_out/r04c/pytest.log
timeout 600 python tools/dev_fuzz.py 40 7071 > gpurun_out/r04c/fuzz.txt 2>&1; tail -n 1 gpurun_out/r04c/fuzz.txt; grep -c " OK " gpurun_out/r04c/fuzz.txt
