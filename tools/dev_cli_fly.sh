#!/bin/bash
# development aid: the CLI on 140 Mb-genome data, index.longread written in sequence (runs 1, 3) and beside the GPU stages (runs 2, 4)
R=$GRAFT_REPO_ROOT; D=/tmp/clifly; mkdir -p $D
$R/tools/hxsim --genome-len ${1:-140000000} --model pacbio --cov 25 --seed 11 --out-prefix $D/s 2>/dev/null
for run in 1 2 3 4; do
  rm -rf $D/out; unset HASLR_INDEX_ASYNC; if [ $((run % 2)) = 0 ]; then export HASLR_INDEX_ASYNC=1; fi
  echo "== run $run"
  t0=$(date +%s.%N)
  $R/haslr_amd/bin/haslr_assemble -t 32 -c $D/s.contigs.fa -l $D/s.reads.fa -m $D/s.paf -d $D/out > $D/log.$run 2>&1
  echo "rc $? wall $(echo "$(date +%s.%N) - $t0" | bc) s"
  grep "NOTE\|elapsed" $D/log.$run | grep -v "number of\|FOFN" | paste - - | cut -c1-200
done
ls -la $D/out/index.* 
