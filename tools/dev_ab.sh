#!/bin/bash
# development: A/B of libhaslr_hip.so builds on the bench data set; args = dir[:ENV=VAL,ENV=VAL] (directories holding a libhaslr_hip.so; "-" = the in-tree build)
cd $GRAFT_REPO_ROOT
run() { python - "$@" <<'PY'
import os, sys, runpy
import haslr_amd.hip as h
if sys.argv[1] != "-":
    h._LIBDIR = os.path.join(os.environ["GRAFT_REPO_ROOT"], sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
PY
}
run - --steps 1 --warmup 0 --no-cpu-baseline --no-configs1 --no-configs3 > /dev/null 2>&1
for rep in 1 2; do
for a in "$@"; do
  d=${a%%:*}; e=""; [ "$a" != "$d" ] && e=$(echo ${a#*:} | tr ',' ' ')
  echo "== $d $e"
  env $e bash -c "$(declare -f run); run $d --steps 4 --warmup 1 --no-cpu-baseline --no-configs3 ${AB_ARGS:---no-configs1}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['ms_per_step'],1), d['poa_phase_cycles']['slowest_edge'], d['assembly']['sha256'][:12], d.get('configs1',{}).get('ms_per_step'))"
done
done
