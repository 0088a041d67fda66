#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03v
HASLR_SKIP_HUGE=1 timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03v/pytest.log 2>&1; tail -3 gpurun_out/r03v/pytest.log
timeout 600 python tools/dev_fuzz.py 50 4041 > gpurun_out/r03v/fuzz.txt 2>&1; tail -1 gpurun_out/r03v/fuzz.txt; grep -c " OK " gpurun_out/r03v/fuzz.txt
HX_DEBUG=1 timeout 600 python bench.py --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/r03v/bench.json 2> gpurun_out/r03v/bench.err
grep "top edge" gpurun_out/r03v/bench.err | tail -5 | cut -c1-170
python -c "
import json;d=json.load(open('gpurun_out/r03v/bench.json'));print('yeast', round(d['ms_per_step'],1), round(d['kernel_ms']['poa'],1), round(d['configs1']['ms_per_step'],1))"
timeout 900 python tools/full_size_check.py fly --no-identity --passes 3 --tmp /tmp/fs > gpurun_out/r03v/fly.json 2> gpurun_out/r03v/fly.err
grep "gpu pass\|oracle_s" gpurun_out/r03v/fly.err; python -c "
import json;d=json.load(open('gpurun_out/r03v/fly.json'));print('fly', d['gcups'], d['parity'])"
