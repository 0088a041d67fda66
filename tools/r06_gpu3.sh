#!/bin/bash
# round 6, GPU call 3: the one-shot binary before / after on the 140 Mb data set, the bench's fly leg with the cold pass (results copied out as they come)
set -u
O=gpurun_out/r06_3
mkdir -p $O
export HASLR_BENCH_DIR=/tmp/haslr_bench
timeout 900 python bench.py --workload fly --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 > $O/bench_fly.json 2> $O/bench_fly.err
cp /tmp/haslr_bench/cli_fly.stderr.txt $O/ 2>/dev/null
FLY=/tmp/haslr_bench/gpu_pacbio_g140000000_s4841534f
cli() { # tag prefix env...
  tag=$1; pre=$2; shift 2
  rm -rf /tmp/cli_$tag
  s=$(date +%s.%N)
  env "$@" HASLR_STAGE_TIMES=$O/cli_$tag.json haslr_amd/bin/haslr_assemble -t 64 -c $pre.contigs.fa -l $pre.reads.fa -m $pre.paf -d /tmp/cli_$tag > /dev/null 2> $O/cli_$tag.err
  e=$(date +%s.%N)
  echo "$tag wall $(python3 -c "print(round($e - $s, 3))") s : $(cat $O/cli_$tag.json)"
  rm -rf /tmp/cli_$tag
}
for i in 1 2; do
  cli fly_old_$i $FLY HASLR_NO_RESERVE=1 HASLR_INDEX_SYNC=1
  cli fly_new_$i $FLY A=1
done > $O/cli_ab.txt 2>&1
HX_DEBUG=1 HASLR_IO_DEBUG=1 haslr_amd/bin/haslr_assemble -t 64 -c $FLY.contigs.fa -l $FLY.reads.fa -m $FLY.paf -d /tmp/cli_dbg > /dev/null 2> $O/cli_fly_debug.err; rm -rf /tmp/cli_dbg
grep -v "hx-edge" $O/cli_fly_debug.err > $O/cli_fly_debug.txt; rm -f $O/cli_fly_debug.err
cat $O/cli_ab.txt
