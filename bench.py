#!/usr/bin/env python
"""bench.py — long-read bases/s through the backbone + consensus path of haslr_assemble on MI355X.

A step is one pass of the hot path over one resident batch of synthetic input:
    chain (filters/sort/dedup/trim/chain, K1-K3) -> edge-support multiset + sort (K4) -> host graph cleaning
    -> edge coordinates (K5) -> POA consensus (K6)
i.e. reference stages main.cpp:115-208: from "inputs parsed and resident" to "all cns_seq
computed". Inputs are uploaded to HBM before the timed region. `value` = long-read bases in the data set
x steps / wall time (max over ranks).

N=1 workload: BASELINE.json configs[1] — E. coli-size 4.6 Mb genome, PacBio-like 25x reads, synthetic
(real E. coli reads cannot be fetched here; tools/hxsim generates contigs + reads + PAF from a seed).
N>1 (weak scaling): genome of N x 4.6 Mb, reads sharded by id range, ONE all-gather of edge records (RCCL),
edges sharded for coordinates + consensus.

Extra objects on the JSON line: `roofline` for the dominant kernel (K6 POA; HBM bound named by the north
star, algorithmic bytes per SURVEY.md 8d) with GCUPS as the secondary figure, and `cpu_baseline` (the
CPU oracle = a port of the reference path, timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before HIP initialises (see haslr_amd/hip.py)

GENOME_PER_GPU = 4_600_000
SEED = 0x4841534C + 1   # seed = 0x4841534c + config index
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_dataset(genome_len, seed, tag):
    d = os.environ.get("HASLR_BENCH_DIR", "/tmp/haslr_bench")
    os.makedirs(d, exist_ok=True)
    pre = os.path.join(d, f"{tag}_g{genome_len}_s{seed:x}")
    if not all(os.path.exists(pre + s) for s in (".contigs.fa", ".reads.fa", ".paf", ".done")):
        sim = os.path.join(ROOT, "tools", "hxsim")
        if not os.path.exists(sim):
            import __graft_entry__
            __graft_entry__.build()
        subprocess.check_call([sim, "--genome-len", str(genome_len), "--seed", hex(seed), "--model", "pacbio", "--cov", "25",
                               "--variant-per-mb", "1.5", "--out-prefix", pre], stderr=subprocess.DEVNULL)
        open(pre + ".done", "w").close()
    return pre


def cpu_baseline(threads):
    """The oracle (kind "port") on a bounded sample of the same workload: a 600 kb genome with the same
    generator settings, all host threads, same timed region."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from haslr_amd import host
    import orclib
    glen = int(os.environ.get("HASLR_BENCH_CPU_GENOME", "600000"))
    pre = make_dataset(glen, SEED, "cpu")
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
    be = orclib.OracleBackend(ds, threads)
    run = host.Run(ds, ds.params(), be.table, None)
    t0 = time.perf_counter()
    run.chain(); run.graph(); run.coords(); run.consensus()
    dt = time.perf_counter() - t0
    st = run.cns_stats()
    out = {"value": ds.total_read_bases / dt, "unit": "long-read bases/s", "cores": threads, "kind": "port",
           "sample": f"synthetic {glen} bp genome, PacBio-like 25x, {ds.reads.n} reads / {ds.total_read_bases} bases, {run.n_edges} edges, "
                     f"{st['dp_cells']} POA cells; oracle/liboracle.so (scalar C++ restatement, {threads} threads over edges) in {dt:.2f} s",
           "gcups": st["dp_cells"] / dt / 1e9}
    run.close(); be.close(); ds.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-len", type=int, default=0, help="override the per-run genome length (testing)")
    ap.add_argument("--poa-block", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from haslr_amd import hip, host

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    torch.zeros(1, device="cuda")   # initialise torch's HIP context (streams, queues) now, not inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    glen = args.genome_len or GENOME_PER_GPU * world
    if rank == 0:
        pre = make_dataset(glen, SEED, "gpu")
    if world > 1:
        dist.barrier()
    pre = make_dataset(glen, SEED, "gpu")
    t0 = time.perf_counter()
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")   # multi-threaded ingest, automatic thread count
    t_parse = time.perf_counter() - t0
    in_bytes = sum(os.path.getsize(pre + x) for x in (".contigs.fa", ".reads.fa", ".paf"))
    prm = ds.params()
    ctx = hip.HipContext(local_rank)
    if args.poa_block:
        ctx.set_poa_block(args.poa_block)
    t1 = time.perf_counter()
    ctx.upload(ds)   # inputs resident in HBM before the timed region
    t_upload = time.perf_counter() - t1
    log(f"[rank {rank}] dataset {pre}: {ds.reads.n} reads, {ds.total_read_bases} bases, {ds.hits.n} PAF records; parse {t_parse:.2f} s ({in_bytes / 1e6 / t_parse:.0f} MB/s), upload {t_upload:.2f} s")

    if world > 1:
        from haslr_amd import distributed as hd
        b = hd.shard_bounds(ds.read_hit_off, ds.reads.n, world)
        ctx.set_read_shard(b[rank], b[rank + 1])
        backend = hd.ShardedBackend(ctx, prm)
        table = backend.table
    else:
        table = ctx.backend()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def step():
        run = host.Run(ds, prm, table, None)
        if world > 1:
            run.set_edge_shard(rank, world)
        run.chain(); run.graph(); run.coords(); run.consensus()
        return run

    for _ in range(args.warmup):
        step().close()
    ctx.timing_reset()
    sync()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        if last is not None:
            last.close()
        ts = time.perf_counter()
        last = step()
        log(f"[rank {rank}] step {time.perf_counter() - ts:.3f} s  stages {last.timings()}")
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-launch figures of the dominant kernel (K6 POA), hipEvent-timed inside the library on its stream
    tim = ctx.timing()
    st = last.cns_stats()
    cns = last.cns_out()
    poa_ms = tim["poa"]["ms"] / max(1, tim["poa"]["launches"])
    alg_bytes = (st["seq_bases"] + 3) // 4 + sum(len(c) for c in cns)   # 2-bit gap bases read once + consensus written once
    stats = torch.tensor([st["dp_cells"], st["seq_bases"], alg_bytes, last.n_edges], dtype=torch.float64, device="cuda")
    tms = torch.tensor([poa_ms], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(stats)
        dist.all_reduce(tms, op=dist.ReduceOp.MAX)
    cells, seq_bases, alg_bytes, n_edges = [float(x) for x in stats.tolist()]
    poa_ms = float(tms.item())
    achieved = alg_bytes / (poa_ms / 1e3) / 1e9 if poa_ms > 0 else 0.0

    # HBM traffic of the POA launch group: measured offline with rocprofv3 PMC passes for exactly this workload (profiles/*_traffic.json)
    traffic, traffic_src = None, None
    if world == 1 and not args.genome_len:
        import glob
        cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
        if cand:
            try:
                traffic = json.load(open(cand[-1]))["hbm_bytes_raw"]
                traffic_src = os.path.relpath(cand[-1], ROOT)
            except Exception:  # noqa: BLE001
                traffic = None
    if rank == 0:
        value = ds.total_read_bases * args.steps / dt
        line = {
            "metric": "long-read bases/sec through backbone+consensus; GFA match + FASTA %identity",
            "value": value, "unit": "long-read bases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"E. coli-size synthetic: {glen} bp genome, PacBio-like 25x long reads + PAF vs short-read contigs "
                                   f"(BASELINE.json configs[1]{' x%d, read-sharded' % world if world > 1 else ''})",
                       "reads": ds.reads.n, "long_read_bases": ds.total_read_bases, "paf_records": ds.hits.n, "edges": int(n_edges),
                       "poa_block_threads": args.poa_block or 256, "parallelism": f"reads+edges sharded x{world}, 1 all-gather" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "k_poa (launch group: one kernel per lane-count class, concurrent)", "kernel_ms_per_launch": poa_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "gcups": cells / (poa_ms / 1e3) / 1e9 if poa_ms > 0 else 0.0, "dp_cells_per_launch": cells,
                         "note": "POA is an O(L^2) integer DP over O(L) bytes: HBM fraction is low by construction (SURVEY.md 8d); GCUPS is the figure of merit"},
            "stage_ms": {k: v * 1e3 for k, v in last.timings().items()},
            "kernel_ms": {k: v["ms"] / max(1, v["launches"]) for k, v in tim.items()},
            "poa_phase_cycles": ctx.poa_phase_cycles(),
            # outside the timed region (the metric starts with parsed, resident inputs): text ingest and the PCIe upload
            "ingest": {"seconds": t_parse, "input_mb": in_bytes / 1e6, "mb_per_s": in_bytes / 1e6 / t_parse, "threads": os.environ.get("HASLR_IO_THREADS", "auto (<= 16)"), "upload_seconds": t_upload},
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(os.cpu_count() or 1)
            except Exception as e:  # noqa: BLE001
                line["cpu_baseline"] = {"error": str(e)}
        print(json.dumps(line), flush=True)
    last.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
