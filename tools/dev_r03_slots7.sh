#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03l
HASLR_SKIP_HUGE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "far_row or persistent or configs1 or fuzz" > gpurun_out/r03l/pytest.log 2>&1
tail -3 gpurun_out/r03l/pytest.log
HX_DEBUG=1 timeout 1200 python tools/full_size_check.py --genome-len 400000000 --model pacbio --name shard400 --no-identity --no-oracle --passes 3 --tmp /tmp/fs > gpurun_out/r03l/s400.json 2> gpurun_out/r03l/s400.err
grep "POA batch" gpurun_out/r03l/s400.err | tail -1 | cut -c1-600; grep "gpu pass" gpurun_out/r03l/s400.err
