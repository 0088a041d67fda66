#!/bin/bash
# development: args = "world[:ENV=VAL,...]" - rank 0's LPT share of the 140 Mb data set for that world size (its first pick is always the costliest edge)
cd $GRAFT_REPO_ROOT
timeout 900 python tools/full_size_check.py fly --no-identity --no-oracle --passes 1 --reuse --tmp /tmp/fs > /dev/null 2>&1
pre=$(ls /tmp/fs/*.paf | head -1 | sed 's/.paf$//')
for a in "$@"; do
  w=${a%%:*}; ee=""; [ "$a" != "$w" ] && ee=$(echo ${a#*:} | tr ',' ' ')
  echo "== world $w $ee"
  env $ee HX_DEBUG=1 python tools/dev_shard_time.py $pre $w 0 2>&1 | grep "^rank\|top edge\|slowest edge" | grep -v "pass 0" | cut -c1-200 | awk '/^rank/{print} /top edge/{c++; if (c<=2) print} /slowest edge/{print}'
done
