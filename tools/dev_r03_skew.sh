#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03d
S=tools/dev_skewbench
{
echo "== local only cm 4 nt 256 1 member"; $S --cols 3000 --len 2000 --members 1 --reps 2 --local 1
for m in 1 4 18; do echo "== remote, members $m"; timeout 60 $S --cols 3000 --len 2000 --members $m --reps 2; done
echo "== remote, members 18 xcd"; timeout 60 $S --cols 3000 --len 2000 --members 18 --reps 2 --xcd 1
for m in 1 8 16 47; do echo "== big remote, members $m"; timeout 60 $S --cols 8000 --len 5000 --members $m --reps 2; done
echo "== big remote, members 47 xcd"; timeout 60 $S --cols 8000 --len 5000 --members 47 --reps 2 --xcd 1
echo "== big remote cm2, members 47"; timeout 60 $S --cols 8000 --len 5000 --members 47 --reps 2 --cm 2
echo "== throughput remote: 64 edges x 18 members"; timeout 60 $S --cols 3000 --len 2000 --members 18 --edges 64 --reps 2
echo "== throughput remote: 2048 edges x 1 member"; timeout 60 $S --cols 1500 --len 1000 --members 1 --edges 2048 --reps 2
echo "== throughput remote: nt 64 8192 edges x 1 member"; timeout 60 $S --cols 1500 --len 1000 --members 1 --edges 8192 --reps 2 --nt 64
} > gpurun_out/r03d/skew.log 2>&1
cat gpurun_out/r03d/skew.log | grep -v "^steps\|rep 0"
