#!/bin/bash
# round 3, first GPU call: skew prototype sweeps, the GPU tests (without the two huge ones), bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03a
S=tools/dev_skewbench
{
echo "== correctness + single member"; $S --cols 3000 --len 2000 --members 1 --reps 2
echo "== members 4"; $S --cols 3000 --len 2000 --members 4 --reps 2
echo "== members 16"; $S --cols 3000 --len 2000 --members 16 --reps 2
echo "== members 16 xcd"; $S --cols 3000 --len 2000 --members 16 --reps 2 --xcd 1
echo "== big: 8000 cols x 5000, members 1"; $S --cols 8000 --len 5000 --members 1 --reps 2
echo "== big: members 16"; $S --cols 8000 --len 5000 --members 16 --reps 2
echo "== big: members 32 xcd"; $S --cols 8000 --len 5000 --members 32 --reps 2 --xcd 1
echo "== big: members 64"; $S --cols 8000 --len 5000 --members 64 --reps 2
echo "== big: members 80 (all bands at once)"; $S --cols 8000 --len 5000 --members 80 --reps 2
echo "== no far edges, members 32"; $S --cols 8000 --len 5000 --members 32 --reps 2 --far 0
echo "== throughput: 64 edges x 4 members"; $S --cols 3000 --len 2000 --members 4 --edges 64 --reps 2
echo "== throughput: 256 edges x 4 members"; $S --cols 3000 --len 2000 --members 4 --edges 256 --reps 2
echo "== throughput: 1024 edges x 1 member"; $S --cols 1500 --len 1000 --members 1 --edges 1024 --reps 2
} > gpurun_out/r03a/skew.log 2>&1
HASLR_SKIP_HUGE=1 timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest.log 2>&1
timeout 900 python bench.py > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err
tail -5 gpurun_out/r03a/pytest.log
cat gpurun_out/r03a/skew.log
