#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03g
for pct in 100 200 400; do
HX_POA_SLOTS_PCT=$pct HX_DEBUG=1 timeout 900 python tools/full_size_check.py fly --no-identity --no-oracle --reuse --passes 3 --tmp /tmp/fs > gpurun_out/r03g/fly_$pct.json 2> gpurun_out/r03g/fly_$pct.err
echo "== pct $pct"; grep "POA batch" gpurun_out/r03g/fly_$pct.err | tail -1 | cut -c1-400; grep "gpu pass" gpurun_out/r03g/fly_$pct.err
done
