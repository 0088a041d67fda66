#!/bin/bash
# development (round 4): one GPU call = a list of steps, each writing under gpurun_out/r04/. Usage: tools/dev_r04.sh step [step ...]
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
fly() { timeout 900 python tools/full_size_check.py fly --no-identity --no-oracle --passes 1 --reuse --tmp /tmp/fs > /dev/null 2>&1; ls /tmp/fs/*.paf | head -1 | sed 's/.paf$//'; }
for step in "$@"; do
  echo "=== $step"
  case $step in
    pkbench) tools/dev_pkbench > $O/pkbench.txt 2>&1; cat $O/pkbench.txt ;;
    bench) timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json ;;
    shard4) pre=$(fly); HX_DEBUG=1 python tools/dev_shard_time.py $pre 4 0 2>&1 | grep "^rank\|top edge\|slowest edge\|class" | grep -v "pass 0" | cut -c1-400 | tee $O/shard4.txt ;;
    shard4v) pre=$(fly); for v in ${SHARD_ENVS:-X=1}; do echo "== $v"; env $(echo $v | tr ',' ' ') HX_DEBUG=1 python tools/dev_shard_time.py $pre ${SHARD_WORLD:-4} 0 2>&1 | grep "^rank\|POA batch:\|wide members\|top edge" | grep -v "pass 0" | cut -c1-420; done | tee $O/shard4v.txt ;;
    flytrace) pre=$(fly); cd /tmp; export TMPDIR=/tmp; for v in ${FLY_ENVS:-X=1}; do echo "== $v"; rm -rf /tmp/ft; env $(echo $v | tr ',' ' ') timeout 900 rocprofv3 --kernel-trace -d /tmp/ft -o kt -- python $GRAFT_REPO_ROOT/tools/full_size_check.py fly --no-identity --no-oracle --passes 2 --reuse --tmp /tmp/fs > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/dev_trace_classes.py /tmp/ft | tail -9; done | tee $GRAFT_REPO_ROOT/$O/flytrace.txt; cd $GRAFT_REPO_ROOT ;;
    flyp1) pre=$(fly); HASLR_DEV_LIBDIR=$GRAFT_REPO_ROOT/haslr_amd/lib_prof1 HX_PROF1=1 HX_DEBUG=1 timeout 900 python tools/full_size_check.py fly --no-identity --no-oracle --passes 2 --reuse --tmp /tmp/fs 2>&1 >/dev/null | grep "prof1" | tail -8 | cut -c1-330 | tee $O/flyp1.txt; HASLR_DEV_LIBDIR=$GRAFT_REPO_ROOT/haslr_amd/lib_prof1 HX_PROF1=1 HX_DEBUG=1 python tools/dev_bench_lib.py haslr_amd/lib_prof1 --steps 1 --warmup 0 --no-cpu-baseline --no-configs1 --no-configs3 2>&1 | grep prof1 | tail -6 | cut -c1-330 | tee -a $O/flyp1.txt ;;
    shard4p3) pre=$(fly); HASLR_DEV_LIBDIR=$GRAFT_REPO_ROOT/haslr_amd/lib_prof3 HX_PROF3=1 HX_DEBUG=1 python tools/dev_shard_time.py $pre 4 0 2>&1 | grep "^rank\|prof3" | cut -c1-400 | tee $O/shard4p3.txt ;;
    shard4p2) pre=$(fly); HASLR_DEV_LIBDIR=$GRAFT_REPO_ROOT/haslr_amd/lib_prof2 HX_PROF2=1 HX_DEBUG=1 python tools/dev_shard_time.py $pre 4 0 2>&1 | grep "^rank\|prof2" | cut -c1-400 | tee $O/shard4p2.txt ;;
    quick) timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_poa_known_answers.py tests/test_spoa_header.py -m gpu -x -q -k "not full_size and not configs and not 65535" 2>&1 | tail -15 | tee $O/quick.txt ;;
    quickv) timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_poa_known_answers.py tests/test_spoa_header.py -m gpu -q -k "not full_size and not configs and not 65535" 2>&1 | grep -v "^       removed" | cut -c1-600 | tail -150 | tee $O/quickv.txt ;;
    quick0) HX_POA_PK16=0 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_poa_known_answers.py tests/test_spoa_header.py -m gpu -q -k "not full_size and not configs and not 65535" 2>&1 | grep -v "^       removed" | cut -c1-600 | tail -60 | tee $O/quick0.txt ;;
    fuzz) timeout 1500 python tools/dev_fuzz.py ${FUZZ_N:-40} ${FUZZ_SEED:-4001} > $O/fuzz.txt 2>&1; tail -n 1 $O/fuzz.txt; grep -c " OK " $O/fuzz.txt; grep -v " OK " $O/fuzz.txt | head -20 ;;
    fuzzbig) FUZZ_BIG=1 timeout 1500 python tools/dev_fuzz.py ${FUZZ_NB:-12} ${FUZZ_SEEDB:-4002} > $O/fuzzbig.txt 2>&1; tail -n 1 $O/fuzzbig.txt; grep -c " OK " $O/fuzzbig.txt; grep -v " OK " $O/fuzzbig.txt | head -20 ;;
    benchq) timeout 900 python bench.py --no-cpu-baseline --no-configs3 > $O/benchq.json 2> $O/benchq.err; python -c "
import json;d=json.load(open('$O/benchq.json'));print('ms/step',d['ms_per_step'],'gcups',d['roofline']['gcups'],'slowest',d['poa_phase_cycles']['slowest_edge'],'configs1',d.get('configs1',{}).get('ms_per_step'))" ;;
    benchq0) HX_POA_PK16=0 timeout 900 python bench.py --no-cpu-baseline --no-configs3 > $O/benchq0.json 2> $O/benchq0.err; python -c "
import json;d=json.load(open('$O/benchq0.json'));print('ms/step',d['ms_per_step'],'gcups',d['roofline']['gcups'],'slowest',d['poa_phase_cycles']['slowest_edge'],'configs1',d.get('configs1',{}).get('ms_per_step'))" ;;
    fly) pre=$(fly); for v in ${FLY_ENVS:-"HX_POA_PK16=1" "HX_POA_PK16=0"}; do echo "== $v"; env $v HX_DEBUG=1 timeout 900 python tools/full_size_check.py fly --no-identity --no-oracle --passes 3 --reuse --tmp /tmp/fs 2>&1 >/dev/null | grep "gpu pass\|POA batch:\|class .: .* workgroups" | tail -12 | cut -c1-300; done | tee $O/fly.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
