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
    shard4p3) pre=$(fly); HASLR_DEV_LIBDIR=$GRAFT_REPO_ROOT/haslr_amd/lib_prof3 HX_PROF3=1 HX_DEBUG=1 python tools/dev_shard_time.py $pre 4 0 2>&1 | grep "^rank\|prof3" | cut -c1-400 | tee $O/shard4p3.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
