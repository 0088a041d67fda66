"""development (round 6): the first consensus call over a freshly reserved arena - fresh context per mode, the 140 Mb data set (bench.py's `fly` workload).
modes: plain | yeast (a 12 Mb context and three passes first) | cli (bench.cli_e2e first) | cli_sleep (... and 5 s) | sleep (2 s between the reserve and the pass)"""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench  # noqa: E402
from haslr_amd import hip, host  # noqa: E402

wl = bench.WORKLOADS[os.environ.get("AB_WORKLOAD", "fly")]
pre = bench.make_dataset(wl, int(wl["genome"]), "gpu")
ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
prm = ds.params()
wy = bench.WORKLOADS["yeast"]
prey = bench.make_dataset(wy, int(wy["genome"]), "gpu")
dsy = host.Dataset(prey + ".contigs.fa", prey + ".reads.fa", prey + ".paf")


def yeast_passes(n):   # a few-edge context first, as in bench.py: other kernel instances on the same streams of the process
    c = hip.HipContext(0)
    c.upload(dsy)
    for _ in range(n):
        r = host.Run(dsy, dsy.params(), c.backend(), None)
        r.chain(); r.graph(); r.coords(); r.consensus()
        r.close()
    c.close()


for mode in sys.argv[1:] or ["yeast", "plain", "cli", "plain"]:
    if mode.startswith("cli"):   # the binary on the same files first, in a process of its own (it reserves 232 GiB and exits): bench.py's order
        bench.cli_e2e(pre, "fly")
        if mode == "cli_sleep":
            time.sleep(5.0)
    if mode.startswith("yeast"):
        yeast_passes(3)
    ctx = hip.HipContext(0)
    r = bench.upload_and_reserve(ctx, ds)
    if mode == "sleep":
        time.sleep(2.0)
    out = []
    for it in range(int(os.environ.get("AB_PASSES", "3"))):
        ctx.timing_reset()
        run = host.Run(ds, prm, ctx.backend(), None)
        t0 = time.perf_counter()
        run.chain(); run.graph(); run.coords(); run.consensus()
        dt = time.perf_counter() - t0
        out.append("%.0f ms (poa kernel %.0f)" % (dt * 1e3, ctx.timing()["poa"]["ms"]))
        print(f"[{mode}] pass {it}: step {dt:.3f} s, poa kernel {ctx.timing()['poa']['ms']:.1f} ms", file=sys.stderr, flush=True)
        run.close()
    print(f"[{mode}] reserve {r['workspace_reserve_s']:.2f} s, arena {ctx.poa_arena_stats()}; passes: " + ", ".join(out), flush=True)
    ctx.close()
