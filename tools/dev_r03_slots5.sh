#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03j
run() { name=$1; shift; env "$@" HX_DEBUG=1 timeout 900 python tools/full_size_check.py fly --no-identity --no-oracle --reuse --passes 3 --tmp /tmp/fs > gpurun_out/r03j/fly_$name.json 2> gpurun_out/r03j/fly_$name.err; echo "== $name"; grep "POA batch" gpurun_out/r03j/fly_$name.err | tail -1 | cut -c1-90; grep "gpu pass" gpurun_out/r03j/fly_$name.err; }
run old HASLR_DEV_LIBDIR=$GRAFT_REPO_ROOT/tools/_oldlib
run new X=1
run cost HX_POA_SLOT_ORDER=cost
run cost200 HX_POA_SLOT_ORDER=cost HX_POA_SLOTS_PCT=200
run old2 HASLR_DEV_LIBDIR=$GRAFT_REPO_ROOT/tools/_oldlib
run new2 X=1
