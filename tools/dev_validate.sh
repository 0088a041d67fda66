#!/bin/bash
# development: the non-huge GPU suite, a fuzz session and a bench line in one GPU call
cd "$GRAFT_REPO_ROOT"
T=${1:-val}
mkdir -p gpurun_out/$T
HASLR_SKIP_HUGE=1 timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/$T/pytest.log 2>&1; tail -n 3 gpurun_out/$T/pytest.log
timeout 900 python tools/dev_fuzz.py ${2:-80} ${3:-6061} > gpurun_out/$T/fuzz.txt 2>&1; tail -n 1 gpurun_out/$T/fuzz.txt; grep -c " OK " gpurun_out/$T/fuzz.txt
FUZZ_BIG=1 timeout 900 python tools/dev_fuzz.py ${4:-20} ${5:-6062} > gpurun_out/$T/fuzz_big.txt 2>&1; tail -n 1 gpurun_out/$T/fuzz_big.txt; grep -c " OK " gpurun_out/$T/fuzz_big.txt
timeout 600 python bench.py --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
python -c "
import json;d=json.loads(open('gpurun_out/$T/bench.json').read().strip().split('\n')[-1]);print('yeast', round(d['ms_per_step'],1), round(d['kernel_ms']['poa'],1), d['poa_phase_cycles']['slowest_edge'], 'ecoli', round(d['configs1']['ms_per_step'],1))"
