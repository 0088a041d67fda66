#!/bin/bash
cd $GRAFT_REPO_ROOT
HX_DEBUG=1 python - tools/_vt --steps 1 --warmup 1 --no-cpu-baseline --no-configs1 <<'PY' 2>&1 | grep "slowest edge\|top edge" | grep -v metric | tail -8 | cut -c1-400
import os, sys, runpy
import haslr_amd.hip as h
h._LIBDIR = os.path.join(os.environ["GRAFT_REPO_ROOT"], sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
PY
