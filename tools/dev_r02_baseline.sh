#!/bin/bash
# round-2 baseline collection on the GPU box: yeast (configs[2]) phase breakdown, kernel trace, SQ counter passes; fly class breakdown
TAG=${1:-r02_base}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
HX_DEBUG=1 timeout 900 python $R/tools/full_size_check.py yeast --no-identity > $O/yeast.json 2> $O/yeast.log
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/tools/full_size_check.py yeast --no-identity --no-oracle > $O/kt.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) > $O/kernel_stats.txt 2>&1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace -d $O/pmc$i -o out -- python $R/tools/full_size_check.py yeast --no-identity --no-oracle --passes 1 > $O/pmc$i.log 2>&1
done
HX_DEBUG=1 timeout 1200 python $R/tools/full_size_check.py fly --no-identity --no-oracle > $O/fly.json 2> $O/fly.log
tail -40 $O/yeast.log
tail -30 $O/fly.log
