#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03h
HASLR_SKIP_HUGE=1 timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03h/pytest.log 2>&1
tail -3 gpurun_out/r03h/pytest.log
timeout 600 python bench.py --no-cpu-baseline --steps 6 > gpurun_out/r03h/bench.json 2> gpurun_out/r03h/bench.err
grep "step " gpurun_out/r03h/bench.err | cut -c1-40
python -c "
import json;d=json.load(open('gpurun_out/r03h/bench.json'));print('yeast', d['value'],d['ms_per_step'],d['roofline']['poa_workspace_bytes'],d['configs1']['ms_per_step'])"
HX_DEBUG=1 timeout 900 python tools/full_size_check.py fly --no-identity --no-oracle --passes 3 --tmp /tmp/fs > gpurun_out/r03h/fly.json 2> gpurun_out/r03h/fly.err
grep "POA batch" gpurun_out/r03h/fly.err | tail -1 | cut -c1-500; grep "gpu pass" gpurun_out/r03h/fly.err
rm -rf /tmp/fs/fly*
HX_DEBUG=1 timeout 1200 python tools/full_size_check.py --genome-len 400000000 --model pacbio --name shard400 --no-identity --no-oracle --passes 2 --tmp /tmp/fs > gpurun_out/r03h/s400.json 2> gpurun_out/r03h/s400.err
grep "POA batch" gpurun_out/r03h/s400.err | cut -c1-500; grep "gpu pass" gpurun_out/r03h/s400.err
