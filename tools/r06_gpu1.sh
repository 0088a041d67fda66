#!/bin/bash
# round 6, GPU call 1: where a one-shot run spends its time (before / after the arena + start-up thread), host time around the POA launch, ingest
# threads, the workspace cap, pruning in the few-edge regime, and the whole GPU suite with per-test durations
set -u
O=gpurun_out/r06_1
mkdir -p $O
export HASLR_BENCH_DIR=/tmp/haslr_bench
{ nproc; free -g; df -h /tmp / /dev/shm; rocm-smi --showmeminfo vram 2>/dev/null | head -8; } > $O/probe.txt 2>&1
timeout 900 python bench.py --steps 6 --warmup 2 > $O/bench.json 2> $O/bench.err
FLY=/tmp/haslr_bench/gpu_pacbio_g140000000_s4841534f
YST=/tmp/haslr_bench/gpu_nanopore_g12000000_s4841534e
ls -la /tmp/haslr_bench > $O/files.txt
cli() { # tag prefix env...
  tag=$1; pre=$2; shift 2
  rm -rf /tmp/cli_$tag
  /usr/bin/time -f "%e s wall %M KB" env "$@" HASLR_STAGE_TIMES=$O/cli_$tag.json haslr_amd/bin/haslr_assemble -t 64 -c $pre.contigs.fa -l $pre.reads.fa -m $pre.paf -d /tmp/cli_$tag > /dev/null 2> $O/cli_$tag.err
  tail -3 $O/cli_$tag.err
  rm -rf /tmp/cli_$tag
}
for i in 1 2; do
  cli fly_old_$i $FLY HASLR_NO_RESERVE=1 HASLR_INDEX_SYNC=1 HX_POA_WORKSPACE_GB=257
  cli fly_new_$i $FLY A=1
  cli fly_t16_$i $FLY A=1
done > $O/cli_ab.txt 2>&1
cli yeast_old $YST HASLR_NO_RESERVE=1 HASLR_INDEX_SYNC=1 >> $O/cli_ab.txt 2>&1
cli yeast_new $YST A=1 >> $O/cli_ab.txt 2>&1
HASLR_IO_DEBUG=1 HX_DEBUG=1 HASLR_GRAPH_DEBUG=1 haslr_amd/bin/haslr_assemble -t 64 -c $FLY.contigs.fa -l $FLY.reads.fa -m $FLY.paf -d /tmp/cli_dbg > /dev/null 2> $O/cli_fly_debug.err; rm -rf /tmp/cli_dbg
python - > $O/ingest_sweep.txt 2>&1 <<PY
import time, sys
sys.path.insert(0, '.')
from haslr_amd import host
pre = "$FLY"
for th in (16, 32, 64, 128, 16):
    t0 = time.perf_counter()
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf", threads=th)
    print(th, "threads", round(time.perf_counter() - t0, 3), "s", flush=True)
    ds.close()
PY
for v in "A=1" "HX_POA_PRUNE=95 HX_POA_PASS_LANES=0" "HX_POA_PRUNE=95 HX_POA_PASS_LANES=0 HX_POA_PRUNE_LANES=512"; do
  echo "== $v"; env $v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs1 --no-configs3 --no-one-shot 2>&1 | grep -E "step |^\{" | cut -c1-400
done > $O/prune_few_edges.txt 2>&1
for v in "A=1" "HX_POA_WORKSPACE_GB=257" "HX_POA_WORKSPACE_GB=100"; do
  echo "== $v"; env $v timeout 300 python bench.py --workload fly --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --no-configs3 --no-one-shot 2>&1 | grep -E "step |^\{" | cut -c1-600
done > $O/workspace_ab.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q --durations=45 > $O/gpu_tests.log 2>&1
tail -5 $O/gpu_tests.log
