#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
HX_DEBUG=1 timeout 1200 python tools/full_size_check.py fly --no-identity --no-oracle --passes 2 --reuse --tmp /tmp/fs > gpurun_out/fly_dbg.json 2> gpurun_out/fly_dbg.err
grep "top edge\|POA batch\|class \|launch\|ms since" gpurun_out/fly_dbg.err | tail -40 | cut -c1-330
