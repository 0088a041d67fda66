#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03n
HASLR_SKIP_HUGE=1 timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03n/pytest.log 2>&1; tail -3 gpurun_out/r03n/pytest.log
timeout 600 python bench.py --no-cpu-baseline --steps 4 > gpurun_out/r03n/bench.json 2> gpurun_out/r03n/bench.err
python -c "
import json;d=json.load(open('gpurun_out/r03n/bench.json'));print('yeast', d['value'],d['ms_per_step'],d['kernel_ms'],d['configs1']['ms_per_step'])"
HASLR_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r03n/bench_2ranks_gloo.json 2> gpurun_out/r03n/bench_2ranks_gloo.err
tail -c 1500 gpurun_out/r03n/bench_2ranks_gloo.json | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('2 ranks', d['value'], d['ms_per_step'], d['assembly'], d['config']['workload'][:80])
except Exception as e: print('2-rank line unreadable', e)"
tail -5 gpurun_out/r03n/bench_2ranks_gloo.err
