#!/bin/bash
# development (round 4, late): one GPU call = the non-huge GPU suite, a bench line, the N > 1 path of bench.py rehearsed (2 ranks over gloo on the one
# GPU; world size 1 over RCCL), and the kernel-trace statistics of the 12 Mb workload
cd "$GRAFT_REPO_ROOT"
T=${1:-r04b}
mkdir -p gpurun_out/$T
if [ "${2:-tests}" = tests ]; then
HASLR_SKIP_HUGE=1 timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/$T/pytest.log 2>&1; tail -n 3 gpurun_out/$T/pytest.log
fi
timeout 600 python bench.py --no-cpu-baseline --no-configs3 --steps 4 --warmup 1 > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
python -c "
import json;d=json.loads(open('gpurun_out/$T/bench.json').read().strip().split('\n')[-1]);print('yeast', round(d['ms_per_step'],1), {k:round(v,2) for k,v in d['kernel_ms'].items()}, d['poa_phase_cycles']['slowest_edge'], 'ecoli', round(d['configs1']['ms_per_step'],1))"
HASLR_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29612 timeout 600 python bench.py --no-cpu-baseline --no-configs1 --no-configs3 --steps 2 --warmup 1 > gpurun_out/$T/bench_nccl_world1.json 2> gpurun_out/$T/bench_nccl_world1.err; tail -c 600 gpurun_out/$T/bench_nccl_world1.json
HASLR_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$T/bench_2ranks_gloo.json 2> gpurun_out/$T/bench_2ranks_gloo.err; tail -c 700 gpurun_out/$T/bench_2ranks_gloo.json
