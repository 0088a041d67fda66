#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03i
for st in 6 8 4; do
HX_POA_STREAMS=$st HX_DEBUG=1 timeout 900 python tools/full_size_check.py fly --no-identity --no-oracle --reuse --passes 3 --tmp /tmp/fs > gpurun_out/r03i/fly_$st.json 2> gpurun_out/r03i/fly_$st.err
echo "== streams $st"; grep "gpu pass" gpurun_out/r03i/fly_$st.err
done
