#!/bin/bash
# development: many-edge regime A/B (140 Mb data set) of libhaslr_hip.so builds; args = dir[:ENV=VAL,ENV=VAL] ("-" = in-tree build)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for a in "$@"; do
  d=${a%%:*}; e=""; [ "$a" != "$d" ] && e=$(echo ${a#*:} | tr ',' ' ')
  echo "== $d $e"
  if [ "$d" = "-" ]; then unset HASLR_DEV_LIBDIR; else export HASLR_DEV_LIBDIR=$GRAFT_REPO_ROOT/$d; fi
  env $e timeout 1200 python tools/full_size_check.py ${FLY_PRESET:-fly} --no-identity --no-oracle --passes 3 --reuse --tmp /tmp/fs > gpurun_out/fly_ab.json 2> gpurun_out/fly_ab.err
  grep "gpu pass" gpurun_out/fly_ab.err | tail -2 | cut -c1-120; python -c "
import json;d=json.load(open('gpurun_out/fly_ab.json'));print('gcups', d.get('gcups'))"
done
