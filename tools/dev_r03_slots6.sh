#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03k
run() { name=$1; shift; env "$@" HX_DEBUG=1 timeout 900 python tools/full_size_check.py fly --no-identity --no-oracle --reuse --passes 3 --tmp /tmp/fs > gpurun_out/r03k/fly_$name.json 2> gpurun_out/r03k/fly_$name.err; echo "== $name"; grep "POA batch" gpurun_out/r03k/fly_$name.err | tail -1 | cut -c1-70; grep "gpu pass" gpurun_out/r03k/fly_$name.err | cut -c1-120; }
run gb260 HX_POA_WORKSPACE_GB=260
run gb140 HX_POA_WORKSPACE_GB=140
run gb70 HX_POA_WORKSPACE_GB=70
run gb35 HX_POA_WORKSPACE_GB=35
run gb260b HX_POA_WORKSPACE_GB=260
