#!/bin/bash
# collects the per-version profile set on the GPU box: GPU tests, an unprofiled bench line, kernel-trace stats, the two PMC traffic passes,
# the SQ counter passes (tools/rocpd_summary.py turns the databases into the summaries committed under profiles/)
TAG=${1:-rXX}
WHAT=${2:-all}      # all | tests | prof | fly (the profile passes on the 140 Mb workload: bench.py --workload fly on the one GPU, written beside the others as fly_*)
BARGS="--no-configs3"; PFX=""
if [ "$WHAT" = fly ]; then BARGS="--workload fly --no-configs3"; PFX="fly_"; fi
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
if [ "$WHAT" = all ] || [ "$WHAT" = tests ]; then
  (cd $R && timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log)
fi
if [ "$WHAT" != fly ]; then
timeout 1500 python $R/bench.py > $O/bench.json 2> $O/bench.log
tail -3 $O/bench.log
fi
if [ "$WHAT" = all ] || [ "$WHAT" = prof ] || [ "$WHAT" = fly ]; then
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/${PFX}kt -o kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs1 --no-one-shot $BARGS > $O/${PFX}bench_under_rocprof.json 2> $O/${PFX}bench_under_rocprof.log
  python $R/tools/rocpd_summary.py $(find $O/${PFX}kt -name "*.db" | head -1) > $O/${PFX}kernel_stats.txt 2>&1
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    timeout 900 rocprofv3 --pmc $grp --kernel-trace -d $O/${PFX}pmc$i -o out -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-configs1 --no-one-shot $BARGS > $O/${PFX}pmc$i.log 2>&1
  done
  python $R/tools/rocpd_summary.py --pmc $O/${PFX}pmc*/ > $O/${PFX}pmc_summary.json 2> $O/${PFX}pmc_summary.log
  cat $O/${PFX}pmc_summary.log | tail -30
  head -12 $O/${PFX}kernel_stats.txt
fi
[ "$WHAT" = fly ] && exit 0
python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().split('\n')[-1])
print({k:d[k] for k in ('value','ms_per_step','n_gpus')}, d['roofline']['gcups'], d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('consensus_equals_gpu'), d.get('configs1'))
"
