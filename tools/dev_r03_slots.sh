#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03f
HASLR_SKIP_HUGE=1 timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03f/pytest.log 2>&1
tail -5 gpurun_out/r03f/pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-configs1 > gpurun_out/r03f/bench.json 2> gpurun_out/r03f/bench.err
python -c "
import json;d=json.load(open('gpurun_out/r03f/bench.json'));print('yeast', d['value'],d['ms_per_step'],d['roofline']['poa_workspace_bytes'],d['poa_phase_cycles']['slowest_edge'])"
HX_DEBUG=1 timeout 900 python tools/full_size_check.py fly --no-identity --tmp /tmp/fs > gpurun_out/r03f/fly.json 2> gpurun_out/r03f/fly.err
grep "POA batch\|gpu pass\|oracle_s" gpurun_out/r03f/fly.err | head -20
python -c "
import json;d=json.load(open('gpurun_out/r03f/fly.json'));print('fly', d['gpu_passes'], d['gcups'], d['parity'])"
