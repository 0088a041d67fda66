#!/bin/bash
# development (round 5): one GPU call = a list of steps, each writing under gpurun_out/r05/. Usage: tools/dev_r05.sh step [step ...]
cd $GRAFT_REPO_ROOT
O=gpurun_out/${R05_OUT:-r05}; mkdir -p $O
for step in "$@"; do
  echo "=== $step"
  case $step in
    quick) HASLR_SKIP_HUGE=1 timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_poa_known_answers.py tests/test_spoa_header.py tests/test_group_sharded.py -m gpu -x -q -k "${QUICK_K:-not eight}" 2>&1 | grep -v "^       removed" | tail -25 | cut -c1-400 | tee $O/quick.txt ;;
    prunetests) HASLR_SKIP_HUGE=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "far_row or persistent or longer_than" 2>&1 | grep -v "^       removed" | tail -25 | cut -c1-400 | tee $O/prunetests.txt ;;
    fuzz) timeout 1500 python tools/dev_fuzz.py ${FUZZ_N:-40} ${FUZZ_SEED:-5001} > $O/fuzz.txt 2>&1; tail -n 1 $O/fuzz.txt; grep -c " OK " $O/fuzz.txt; grep -v " OK " $O/fuzz.txt | head -20 ;;
    fuzzbig) FUZZ_BIG=1 timeout 1500 python tools/dev_fuzz.py ${FUZZ_NB:-12} ${FUZZ_SEEDB:-5002} > $O/fuzzbig.txt 2>&1; tail -n 1 $O/fuzzbig.txt; grep -c " OK " $O/fuzzbig.txt; grep -v " OK " $O/fuzzbig.txt | head -20 ;;
    ab140) timeout 1500 python tools/dev_r05_ab.py ${AB_SPECS:-"poa_prune=0" "-" "poa_prune=90" "poa_prune=100"} 2> $O/ab140.err | tee $O/ab140.txt; grep -c "removed" $O/ab140.err ;;
    ab140dbg) HX_DEBUG=1 AB_PASSES=2 timeout 1500 python tools/dev_r05_ab.py ${AB_SPECS:-"-"} 2>&1 | grep "RESULT\|pruning\|class .: .* workgroups\|POA batch:\|top edge" | cut -c1-420 | tee $O/ab140dbg.txt ;;
    edgedump) HX_DEBUG=2 AB_PASSES=2 timeout 1500 python tools/dev_r05_ab.py ${AB_SPECS:-"-"} 2> $O/edgedump.err | tee $O/edgedump.txt; grep -c "hx-edge" $O/edgedump.err ;;
    ab12) AB_WORKLOAD=yeast timeout 900 python tools/dev_r05_ab.py ${AB12_SPECS:-"-" "poa_prune=95"} 2> $O/ab12.err | tee $O/ab12.txt ;;
    benchq) timeout 1500 python bench.py --no-cpu-baseline --steps ${BENCH_STEPS:-6} --warmup 2 > $O/benchq.json 2> $O/benchq.err; python -c "
import json;d=json.load(open('$O/benchq.json'));print('ms/step',d['ms_per_step'],'gcups',d['roofline']['gcups'],'slowest',d['poa_phase_cycles']['slowest_edge'],'config',d['config'])" ;;
    bench) timeout 1800 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 2500 $O/bench.json ;;
    alltests) timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -v "^       removed" | tail -30 | cut -c1-400 | tee $O/alltests.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
