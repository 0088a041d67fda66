"""development: consensus time of ONE rank's edge share (LPT over `world` ranks) of a full-size data set on one GPU - what a rank of an N-GPU run
has to do, measured without the other GPUs (data set of tools/full_size_check.py <preset> --reuse --tmp /tmp/fs)"""
import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
from haslr_amd import hip, host
if os.environ.get('HASLR_DEV_LIBDIR'):   # (development: another build of libhaslr_hip.so)
    hip._LIBDIR = os.environ['HASLR_DEV_LIBDIR']
pre, world = sys.argv[1], int(sys.argv[2])
ranks = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else list(range(world))
ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf", threads=min(32, os.cpu_count() or 1))
ctx = hip.HipContext(0)
ctx.upload(ds)
for rank in ranks:
    for it in range(2):
        run = host.Run(ds, ds.params(), ctx.backend(), None)
        run.set_edge_shard(rank, world)
        run.chain(); run.graph(); run.coords()
        t0 = time.perf_counter(); run.consensus(); dt = time.perf_counter() - t0
        print("rank", rank, "of", world, "pass", it, "edges", run.n_edges, "consensus %.3f s" % dt, "cells %.3g" % run.cns_stats()["dp_cells"], flush=True)
        if os.environ.get("HX_DEBUG") and it == 1:
            ctx.poa_phase_cycles()
        run.close()
