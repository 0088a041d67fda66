#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python tools/full_size_check.py fly --no-identity --no-oracle --passes 1 --reuse --tmp /tmp/fs > /dev/null 2>&1
pre=$(ls /tmp/fs/*.paf | head -1 | sed 's/.paf$//')
for e in "$@"; do
  ee=""; [ "$e" != "-" ] && ee=$(echo $e | tr ',' ' ')
  echo "== $ee"
  env $ee HX_DEBUG=1 python tools/dev_shard_time.py $pre 4 0,1 2>&1 | grep "^rank\|top edge\|wide members" | grep -v "pass 0" | cut -c1-200 | head -16
done
