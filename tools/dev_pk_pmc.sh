cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in 0 1; do
  rm -rf /tmp/pp$v
  HX_POA_PK16=$v timeout 900 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d /tmp/pp$v -o out -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-configs1 --workload fly --no-configs3 > /tmp/pp$v.log 2>&1
  echo "== PK16=$v"; python $R/tools/rocpd_summary.py --pmc /tmp/pp$v/ 2>&1 >/dev/null | tail -12 | cut -c1-220
done
