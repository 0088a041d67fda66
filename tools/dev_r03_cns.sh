#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03u
HASLR_SKIP_HUGE=1 timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03u/pytest.log 2>&1; tail -3 gpurun_out/r03u/pytest.log
HX_DEBUG=1 timeout 600 python bench.py --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/r03u/bench.json 2> gpurun_out/r03u/bench.err
grep "top edge" gpurun_out/r03u/bench.err | tail -5 | cut -c1-170
python -c "
import json;d=json.load(open('gpurun_out/r03u/bench.json'));print('yeast', round(d['ms_per_step'],1), round(d['kernel_ms']['poa'],1), d['poa_phase_cycles']['slowest_edge'], round(d['configs1']['ms_per_step'],1))"
