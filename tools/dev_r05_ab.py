"""development (round 5): the 140 Mb data set (bench.py's `fly` workload) on the one GPU under several option sets in ONE process - step time, kernel
time, pruning statistics and the consensus digest of each (they must all be equal). Usage: dev_r05_ab.py "k=v,k=v" "k=v" ...  ("-" = defaults)"""
import hashlib
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench  # noqa: E402
from haslr_amd import hip, host  # noqa: E402

if os.environ.get("AB_LIBDIR"):   # a variant build of libhaslr_hip.so (tools/dev_variant.sh)
    hip._LIBDIR = os.path.join(ROOT, os.environ["AB_LIBDIR"])

wl_name = os.environ.get("AB_WORKLOAD", "fly")
wl = bench.WORKLOADS[wl_name]
glen = int(os.environ.get("AB_GENOME", wl["genome"]))
pre = bench.make_dataset(wl, glen, "gpu")
ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
ctx = hip.HipContext(0)
ctx.upload(ds)
prm = ds.params()
passes = int(os.environ.get("AB_PASSES", "3"))
for spec in sys.argv[1:] or ["-"]:
    opts = dict(kv.split("=") for kv in spec.split(",")) if spec != "-" else {}
    with ctx.options(**opts):
        best = None
        for it in range(passes):
            ctx.timing_reset()
            run = host.Run(ds, prm, ctx.backend(), None)
            t0 = time.perf_counter()
            run.chain(); run.graph(); run.coords(); run.consensus()
            dt = time.perf_counter() - t0
            tm = ctx.timing()
            if it == passes - 1:
                h = hashlib.sha256()
                for c in run.cns_out():
                    h.update(c if isinstance(c, bytes) else str(c).encode()); h.update(b"\n")
                st = run.cns_stats()
                pr = ctx.poa_prune_stats()
                ph = ctx.poa_phase_cycles()
                mem = ctx.poa_memory_stats()
            best = dt if best is None else min(best, dt)
            phq = ctx.poa_phase_cycles() if os.environ.get("AB_PHASES") else None
            parts = "" if phq is None else ", ".join("%s %.1f" % (k, v / 2.4e6) for k, v in phq["slowest_edge"].items())
            extra = "" if phq is None else f", slowest edge {sum(phq['slowest_edge'].values()) / 2.4e6:.1f} ms ({parts}), all edges {sum(phq['sum'].values()) / 1e9:.1f} Gcycles"
            print(f"[{spec}] pass {it}: step {dt:.3f} s, poa kernel {tm['poa']['ms']:.1f} ms{extra}", flush=True)
            run.close()
        skipped = pr["wave_rows_skipped"] / pr["wave_rows"] if pr["wave_rows"] else 0.0
        print(f"RESULT [{spec}] best step {best:.3f} s | edges {ph['edges']} cells {st['dp_cells']:.4g} gcups {st['dp_cells'] / best / 1e9:.0f} | pruned wave-rows {pr['wave_rows']:.4g} skipped {skipped:.3f} "
              f"thresholds {pr['alignments_with_threshold']} repeated {pr['attempts_repeated']} | workspace {mem['last_call_workspace'] / 1e9:.1f} GB | slowest edge {sum(ph['slowest_edge'].values()) / 2.4e6:.1f} ms | consensus {h.hexdigest()[:16]}", flush=True)
ctx.close()
