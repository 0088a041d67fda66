"""development (round 6): the 12 Mb step in the first and in a second context of one process (the second used to be 13 % slower)"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench
from haslr_amd import hip, host
wl = bench.WORKLOADS["yeast"]
pre = bench.make_dataset(wl, wl["genome"], "gpu")
ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
prm = ds.params()
for k in range(3):
    ctx = hip.HipContext(0)
    ctx.upload(ds)
    best = 1e9
    for it in range(5):
        run = host.Run(ds, prm, ctx.backend(), None)
        t0 = time.perf_counter()
        run.chain(); run.graph(); run.coords(); run.consensus()
        best = min(best, time.perf_counter() - t0)
        run.close()
    print(f"context {k}: best step {best * 1e3:.1f} ms", flush=True)
    ctx.close()
