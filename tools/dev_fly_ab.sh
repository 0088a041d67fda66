#!/bin/bash
# development: many-edge regime A/B (140 Mb data set) of libhaslr_hip.so builds; args = directories ("-" = in-tree)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for d in "$@"; do
  echo "== $d"
  if [ "$d" = "-" ]; then unset HASLR_DEV_LIBDIR; else export HASLR_DEV_LIBDIR=$GRAFT_REPO_ROOT/$d; fi
  timeout 1200 python tools/full_size_check.py fly --no-identity --no-oracle --passes 3 --reuse --tmp /tmp/fs > gpurun_out/fly_ab.json 2> gpurun_out/fly_ab.err
  grep "gpu pass" gpurun_out/fly_ab.err | tail -3; python -c "
import json;d=json.load(open('gpurun_out/fly_ab.json'));print('fly gcups', d.get('gcups'))"
done
