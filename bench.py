#!/usr/bin/env python
"""bench.py — long-read bases/s through the backbone + consensus path of haslr_assemble on MI355X.

A step is one pass of the hot path over one resident batch of synthetic input:
    chain (filters/sort/dedup/trim/chain, K1-K3) -> edge-support multiset + sort (K4) -> host graph cleaning
    -> edge coordinates (K5) -> POA consensus (K6) [-> N>1: all-gather of the per-edge results]
i.e. reference stages main.cpp:115-208: from "inputs parsed and resident" to "all cns_seq
computed" (on every rank, when there are several). Inputs are uploaded to HBM before the timed region.
`value` = long-read bases in the data set x steps / wall time (max over ranks).

N=1 workload: BASELINE.json configs[2], the largest single-GPU configuration — S. cerevisiae-size 12 Mb genome,
Nanopore-like 25x long reads + PAF against short-read contigs, synthetic (tools/hxsim, seed 0x4841534c + 2;
there is no network for real reads). configs[1] (E. coli-size 4.6 Mb, PacBio-like 25x) is measured too and
reported under "configs1" (`--workload ecoli` makes it the main line instead); configs[3]'s 140 Mb data set is run on
the one GPU as well and reported under "configs3" (the many-edge regime: throughput, not one edge's chain).
N>1: reads sharded by id range, ONE all-gather of edge records (RCCL), edges sharded by estimated DP cost for
coordinates + consensus, one all-gather of the results; the ranks agree on success before every collective.
N = 4 runs BASELINE.json configs[3] as it is named (140 Mb PacBio-like, read-sharded over 4 GPUs); N = 2 and N = 8 give every GPU a 140 Mb
chromosome's worth (280 Mb; 1.12 Gb = the shape of configs[4] at 0.36 x its size): the many-edge regime the north star's target lives in
(`config.workload` says which). Launched WITHOUT torch.distributed.run, `--gpus N` runs the product binary's own multi-GPU path instead: N ranks
inside this process (hx_group_create, one RCCL all-gather in hx_edge_merge, hxh_runs_all_sharded - what `haslr_assemble --gpus N` runs). After the timed steps every rank stitches the
assembly, the ranks' assemblies must be identical, and rank 0 repeats the pass on its GPU alone and requires the
same assembly (`assembly.matches_single_gpu`).

Extra objects on the JSON line: `roofline` for the dominant kernel (K6 POA; HBM bound named by the north
star, algorithmic bytes per SURVEY.md 8d) with GCUPS against the VALU issue bound (256 CU x 4 SIMD-32 x 2.4 GHz),
the SQ-counter figures of the committed profile, the critical path and the POA workspace; `gfa` +
`gfa_equals_cpu_baseline` (the metric's "GFA match": six files, byte for byte); and `cpu_baseline` (the
CPU oracle = a port of the reference path with AVX2 row kernels, timed on this box's host cores on the SAME
data set, 64 threads and all cores; its consensus must equal the GPU's).
"""
import argparse
import glob
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before HIP initialises (see haslr_amd/hip.py)

SEED0 = 0x4841534C                                # seed = 0x4841534c + config index
WORKLOADS = {
    "yeast": dict(config=2, genome=12_000_000, model="nanopore", name="S. cerevisiae-size synthetic", variants="1.5"),
    "ecoli": dict(config=1, genome=4_600_000, model="pacbio", name="E. coli-size synthetic", variants="1.5"),
    "fly": dict(config=3, genome=140_000_000, model="pacbio", name="D. melanogaster-size synthetic", variants="1.5"),
}
HBM_PEAK_GBS = 8000.0                             # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9         # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 (a wave64 VALU op issues in 2 cycles) x 2.4 GHz = 7.86e13 lane-ops/s
SHADER_CLOCK_HZ = 2.4e9                           # clock the kernel's cycle counters are converted with (critical_path_ms)
VALU_OPS_PER_CELL_MODEL = 10                      # ~10 lane-ops per DP cell is the model the VALU bound is quoted for


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_dataset(wl, genome_len, tag, chromosomes=1):
    """chromosomes > 1: the genome as that many independent chromosomes, simulated in parallel (tools/hxsim --chromosomes: chromosome k with seed + 7919 k)"""
    d = os.environ.get("HASLR_BENCH_DIR", "/tmp/haslr_bench")
    os.makedirs(d, exist_ok=True)
    seed = SEED0 + wl["config"]
    pre = os.path.join(d, f"{tag}_{wl['model']}_g{genome_len}_s{seed:x}" + (f"_c{chromosomes}" if chromosomes > 1 else ""))
    if not all(os.path.exists(pre + s) for s in (".contigs.fa", ".reads.fa", ".paf", ".done")):
        sim = os.path.join(ROOT, "tools", "hxsim")
        if not os.path.exists(sim):
            import __graft_entry__
            __graft_entry__.build()
        extra = ["--chromosomes", str(chromosomes)] if chromosomes > 1 else []
        subprocess.check_call([sim, "--genome-len", str(genome_len), "--seed", hex(seed), "--model", wl["model"], "--cov", "25",
                               "--variant-per-mb", wl["variants"], "--out-prefix", pre] + extra, stderr=subprocess.DEVNULL)
        open(pre + ".done", "w").close()
    return pre


def multi_gpu_workload(world, workload=None, genome_len=0):
    """What N > 1 GPUs run by default - the MANY-EDGE regime the north star's target lives in (thousands of edges per GPU: throughput, not one edge's
    chain): N = 4 is BASELINE configs[3] as named (140 Mb read-sharded over 4 GPUs, fixed size); N = 2 and N = 8 give every GPU a 140 Mb chromosome's
    worth (280 Mb / 1.12 Gb: weak scaling; N = 8 = the shape of configs[4] - CHM1, 3.1 Gb on 8 GPUs - at 0.36 x its size, which is what fits the
    driver's run time: ~1 min to simulate in parallel, ~70 GB of text). Returns (workload key, genome length, chromosomes, as_named)."""
    if workload is None:
        workload = "fly" if world > 1 else "yeast"
    wl = WORKLOADS[workload]
    as_named = workload == "fly" and world == 4
    if genome_len:
        return workload, genome_len, max(1, world if genome_len >= 2 * 17_000_000 else 1), False
    if world == 1 or as_named:
        return workload, wl["genome"], 1, as_named
    return workload, wl["genome"] * world, world if workload == "fly" else 1, False


def gfa_digest(out_dir):
    """sha256 of each of the six backbone.0x.*.gfa files of a run directory (the metric's "GFA match")"""
    files = sorted(glob.glob(os.path.join(out_dir, "backbone.0[1-6].*.gfa")))
    return {os.path.basename(f): hashlib.sha256(open(f, "rb").read()).hexdigest() for f in files}


def bench_dir():
    d = os.environ.get("HASLR_BENCH_DIR", "/tmp/haslr_bench")
    os.makedirs(d, exist_ok=True)
    return d


def cli_e2e(pre, tag):
    """ONE run of the drop-in binary on the text files, in a process of its own: what a user of bin/haslr.py:66 waits for (the reference is a run-once
    program, main.cpp:28-228). Wall time of the process + the binary's own account of it (HASLR_STAGE_TIMES). The input files were just written by
    the simulator (page cache warm); index.contig / index.longread are written as in any first run; -t = min(64, cores)."""
    import shutil
    exe = os.path.join(ROOT, "haslr_amd", "bin", "haslr_assemble")
    out = os.path.join(bench_dir(), f"cli_{tag}")
    shutil.rmtree(out, ignore_errors=True)
    times = os.path.join(bench_dir(), f"cli_{tag}.times.json")
    threads = min(64, os.cpu_count() or 1)
    env = dict(os.environ, HASLR_STAGE_TIMES=times)
    env.pop("HX_DEBUG", None)
    t0 = time.perf_counter()
    r = subprocess.run([exe, "-t", str(threads), "-c", pre + ".contigs.fa", "-l", pre + ".reads.fa", "-m", pre + ".paf", "-d", out], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    wall = time.perf_counter() - t0
    open(os.path.join(bench_dir(), f"cli_{tag}.stderr.txt"), "w").write(r.stderr)
    if r.returncode != 0:
        return {"error": r.stderr[-600:]}
    res = {"cli_e2e_s": wall, "threads": threads, "command": "haslr_assemble -t %d -c .. -l .. -m .. -d .. (fresh process, fresh output directory)" % threads}
    try:
        res["stages_s"] = json.load(open(times))
    except Exception as e:  # noqa: BLE001
        res["stages_s"] = {"error": str(e)}
    try:
        res["asm_final_fa_sha256"] = hashlib.sha256(open(os.path.join(out, "asm.final.fa"), "rb").read()).hexdigest()
        res["gfa"] = gfa_digest(out)
    except Exception as e:  # noqa: BLE001
        res["asm_final_fa_sha256"] = "error: " + str(e)
    shutil.rmtree(out, ignore_errors=True)
    return res


def reserve_bytes_for(ds):
    """the arena haslr_assemble reserves beside its parse: 64 B of consensus workspace per long-read base, at most what a call ever settles on (main.cpp of the build)"""
    return min(232 << 30, 64 * int(ds.total_read_bases))


def upload_and_reserve(ctx, ds):
    """inputs to HBM while a second thread reserves the consensus workspace's arena - the start-up of the product binary (hx_poa_reserve beside the
    parse / upload). HASLR_BENCH_NO_RESERVE=1: no reservation (the first consensus call allocates, as before round 6)."""
    import threading
    t0 = time.perf_counter()
    th, t_res = None, [0.0]
    if os.environ.get("HASLR_BENCH_NO_RESERVE") != "1":
        def work():
            a = time.perf_counter()
            ctx.poa_reserve(reserve_bytes_for(ds))
            t_res[0] = time.perf_counter() - a
        th = threading.Thread(target=work)
        th.start()
    ctx.upload(ds)
    t_up = time.perf_counter() - t0
    if th:
        th.join()
    return {"upload_s": t_up, "workspace_reserve_s": t_res[0], "upload_and_reserve_s": time.perf_counter() - t0, "reserved_bytes": ctx.poa_arena_stats()["bytes"]}


def cold_pass(ctx, ds, prm, table):
    """the FIRST pass of a fresh context over a data set (untimed for `value`; what a one-shot program pays): every scratch array is allocated, every
    kernel runs for the first time. The consensus workspace's arena was reserved beside the upload (upload_and_reserve), as the binary does."""
    from haslr_amd import host
    t0 = time.perf_counter()
    run = host.Run(ds, prm, table, None)
    run.chain(); run.graph(); run.coords(); run.consensus()
    dt = time.perf_counter() - t0
    tm = run.timings()
    res = {"cold_first_pass_ms": dt * 1e3, "cold_consensus_ms": tm.get("consensus", 0) * 1e3, "cold_stage_ms": {k: v * 1e3 for k, v in tm.items()},
           "cold_poa_host_ms": ctx.poa_host_times(), "workspace_arena": ctx.poa_arena_stats(), "poa_workspace_bytes": ctx.poa_workspace_bytes()}
    run.close()
    return res


def cpu_baseline(ds, gpu_cns, gpu_gfa=None, work_dir=None):
    """The oracle (kind "port": the CPU restatement of the reference path, AVX2 row kernels where the host has them,
    edges dealt to the threads costliest first) on the SAME data set and timed region as the GPU line, twice:
    min(64, cores) threads - the thread count the north star quotes the reference at - and all cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from haslr_amd import host
    import orclib
    ncpu = os.cpu_count() or 1
    runs = []
    for threads in sorted({min(64, ncpu), ncpu}):
        be = orclib.OracleBackend(ds, threads)
        run = host.Run(ds, ds.params(), be.table, None)
        t0 = time.perf_counter()
        run.chain(); run.graph(); run.coords(); run.consensus()
        dt = time.perf_counter() - t0
        st = run.cns_stats()
        runs.append({"threads": threads, "seconds": dt, "value": ds.total_read_bases / dt, "gcups": st["dp_cells"] / dt / 1e9,
                     "stage_s": run.timings(), "consensus_equals_gpu": run.cns_out() == gpu_cns})
        n_edges = run.n_edges
        run.close()
        if gpu_gfa is not None and "gfa" not in runs[0]:   # untimed: chain + graph once more with an output directory, for the six GFA snapshots
            d = os.path.join(work_dir, "cpu_gfa")
            r2 = host.Run(ds, ds.params(), be.table, d)
            r2.chain(); r2.graph(); r2.close()
            runs[0]["gfa"] = gfa_digest(d)
        be.close()
    main = runs[0]
    gfa_equal = None if gpu_gfa is None else (len(gpu_gfa) == 6 and main.get("gfa") == gpu_gfa)
    return {"value": main["value"], "unit": "long-read bases/s", "cores": main["threads"], "kind": "port",
            "sample": f"the whole data set of this line ({ds.reads.n} reads / {ds.total_read_bases} bases, {n_edges} edges), same timed region "
                      f"(chain -> graph -> coordinates -> consensus); oracle/liboracle.so, POA row kernels {orclib.lib().orc_poa_kernel_name().decode()}, "
                      f"{main['threads']} threads over edges (costliest first) in {main['seconds']:.2f} s",
            "gcups": main["gcups"], "consensus_equals_gpu": all(r["consensus_equals_gpu"] for r in runs), "gfa_equals_gpu": gfa_equal, "host_cores": ncpu, "runs": runs,
            "note": "stand-in for '64-thread CPU haslr_assemble' (the reference cannot be built here: spoa 1.1.3 is not vendored); AVX2 rows, 16 x int16 where "
                    "8 x (nodes + columns) fits 16 bits (the dispatch of spoa's SIMD engine), 8 x int32 elsewhere"}


def measure(ctx, ds, prm, table, steps, warmup, world, rank, sync, gather, lr_begin, sharded=None):
    """`warmup` untimed + `steps` timed passes; returns (seconds, last run)."""
    from haslr_amd import host

    if sharded is None:
        sharded = world > 1

    def step():
        run = host.Run(ds, prm, table, None)
        if sharded:
            run.set_edge_shard(rank, world)
            run.set_read_shard(lr_begin)
        if sharded:
            gather(run)                            # chain -> merged graph -> coords -> consensus -> results gathered; the ranks agree on success before every collective
        else:
            run.chain(); run.graph(); run.coords(); run.consensus()
        return run

    for _ in range(warmup):
        step().close()
    ctx.timing_reset()
    sync()
    t0 = time.perf_counter()
    last, lines = None, []
    for _ in range(steps):
        if last is not None:
            last.close()      # (inside the timed region on purpose: the next pass's result arrays then reuse these pages - kept to the end instead, every pass faults its 10 MB in anew and takes 1 ms longer)
        ts = time.perf_counter()
        last = step()
        lines.append(f"[rank {rank}] step {time.perf_counter() - ts:.3f} s  stages {last.timings()}")
    sync()
    dt = time.perf_counter() - t0
    for line in lines:
        log(line)
    return dt, last


def cpu_sample(ds, share=8):
    """CPU oracle on a BOUNDED sample of the data set of a multi-GPU line: chain + graph over everything (seconds), coordinates + consensus over the LPT
    share 1 / `share` of the edges; the whole-data-set rate is extrapolated from it (consensus is > 95 % of the CPU pass) and says so."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from haslr_amd import host
    import orclib
    threads = min(64, os.cpu_count() or 1)
    be = orclib.OracleBackend(ds, threads)
    run = host.Run(ds, ds.params(), be.table, None)
    run.set_edge_shard(0, share)
    t0 = time.perf_counter()
    run.chain(); run.graph()
    t_front = time.perf_counter() - t0
    t0 = time.perf_counter()
    run.coords(); run.consensus()
    t_share = time.perf_counter() - t0
    n_share, n_all = run.n_edges, run.n_edges_total
    cells = run.cns_stats()["dp_cells"]
    run.close(); be.close()
    est = t_front + t_share * share
    return {"value": ds.total_read_bases / est, "unit": "long-read bases/s", "cores": threads, "kind": "port", "extrapolated": True,
            "sample": f"chain + graph over the whole data set ({t_front:.2f} s), coordinates + consensus over the LPT share 1/{share} of the edges ({n_share} of {n_all}, "
                      f"{cells:.3g} cells, {t_share:.2f} s); whole-data-set time estimated as {t_front:.2f} + {share} x {t_share:.2f} = {est:.1f} s; oracle/liboracle.so, {threads} threads",
            "gcups_on_sample": cells / t_share / 1e9, "see": "the N = 1 line carries the CPU leg on a whole data set with the consensus compared"}


def run_group(args):
    """`--gpus N` without a launcher: the product binary's multi-GPU path driven from here - N ranks INSIDE this process (hx_group_create: one context +
    one RCCL communicator per rank; HASLR_GROUP_TRANSPORT=host stages the exchange through host memory and lets the ranks share a device: the
    rehearsal on a one-GPU box), hxh_runs_all_sharded = one host thread per rank: chain (own reads) -> hx_edge_merge (ONE ncclAllGather of the packed
    records) -> cleaning (redundant) -> coordinates + consensus (own share of the queue, LPT) -> results through the process's memory -> rank 0
    assembles. A step is timed from its start to the end of the consensus stage (the assembly that the call also makes is outside the metric).
    The line carries what the N = 1 line carries: `roofline` (algorithmic bytes of all ranks over the slowest rank's POA launch group), `cpu_baseline`
    (a bounded sample, extrapolated), and an N = 1 pass of the plain path over the same data on rank 0's device (`n1_same_data`)."""
    import ctypes as C

    import torch
    from haslr_amd import ctypes_defs as T
    from haslr_amd import hip, host
    if torch.cuda.is_available():
        torch.zeros(1, device="cuda")   # torch's HIP context first, as in the launcher path (it bundles its own runtime)
        torch.cuda.synchronize()
    n = args.gpus
    workload, glen, chroms, as_named = multi_gpu_workload(n, args.workload, args.genome_len)
    wl = WORKLOADS[workload]
    pre = make_dataset(wl, glen, "gpu", chroms)
    t0 = time.perf_counter()
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
    t_parse = time.perf_counter() - t0
    prm = ds.params()
    L, H = hip.lib(), host.lib()
    def sync():
        torch.cuda.synchronize()

    def solo_reference(warm):
        """the plain single-context path (bench.py's N = 1 step) over the same data on device 0: the sharded assembly must equal its assembly, its steady step is this line's N = 1 reference"""
        ctx = hip.HipContext(0)
        upload_and_reserve(ctx, ds)
        solo = host.Run(ds, prm, ctx.backend(), None)
        t0 = time.perf_counter()
        solo.all()
        t_solo = time.perf_counter() - t0
        sha1 = hashlib.sha256(solo.assembly_fasta().encode()).hexdigest()
        solo.close()
        dt1, run1 = measure(ctx, ds, prm, ctx.backend(), args.steps, warm, 1, 0, sync, None, 0, sharded=False)
        run1.close(); ctx.close()
        return sha1, t_solo, {"value": ds.total_read_bases * args.steps / dt1, "ms_per_step": dt1 / args.steps * 1e3, "steps": args.steps}
    # One rank (the group code at N = 1, HASLR_BENCH_FORCE_GROUP): the N = 1 reference is the plain bench.py line of A PROCESS OF ITS OWN over the same files, so that
    # neither pass inherits anything of the other's (until the launch classes' streams became the process's - hx_ctx_create - whichever context was created second in a
    # process ran the 12 Mb step 13 % slower: plain 184.6 ms after the group's 162.1 ms, group 172-219 ms after the plain pass's 153 ms). Several ranks: in this process,
    # afterwards, when the group has given its memory back (a reference for the assembly and a rough N = 1 figure).
    solo_first = n == 1
    if solo_first:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(max(1, args.warmup)), "--no-cpu-baseline", "--no-configs1", "--no-configs3", "--no-one-shot"]
        if args.workload:
            cmd += ["--workload", args.workload]
        if args.genome_len:
            cmd += ["--genome-len", str(args.genome_len)]
        env = {k: v for k, v in os.environ.items() if k != "HASLR_BENCH_FORCE_GROUP"}
        t0 = time.perf_counter()
        pr = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if pr.returncode != 0:
            raise SystemExit("bench.py (plain N = 1 reference in a process of its own): " + pr.stderr[-800:])
        ref = json.loads(pr.stdout.strip().split("\n")[-1])
        solo_sha, t_solo = ref["assembly"]["sha256"], time.perf_counter() - t0
        n1 = {"value": ref["value"], "ms_per_step": ref["ms_per_step"], "steps": ref["steps"],
              "what": "the plain bench.py line (N = 1 step, single context) of a process of its own over the same files, run before the group was created"}
    g = C.c_void_p()
    tr = os.environ.get("HASLR_GROUP_TRANSPORT")
    if L.hx_group_create(n, None, tr.encode() if tr else None, C.byref(g)) != 0:
        raise SystemExit("bench.py --gpus %d (in-process group): %s" % (n, L.hx_last_error().decode()))
    transport = L.hx_group_transport(g).decode()
    rr = (C.c_int * n)()
    L.hx_group_rccl_ranks(g, rr)
    bounds = host.shard_bounds(ds, n)
    tables = [T.Backend() for _ in range(n)]
    ctxs = [L.hx_group_ctx(g, r) for r in range(n)]
    t1 = time.perf_counter()
    for r in range(n):
        c = ctxs[r]
        for k, v in hip.env_options().items():
            if L.hx_set_option(c, k.encode(), str(v).encode()) != 0:
                raise SystemExit(L.hx_last_error().decode())
        if L.hx_upload(c, C.byref(ds.contigs), C.byref(ds.reads), C.byref(ds.hits), ds.read_hit_off) != 0 or L.hx_set_read_shard(c, bounds[r], bounds[r + 1]) != 0:
            raise SystemExit(L.hx_last_error().decode())
        L.hx_group_backend_fill(g, r, C.byref(tables[r]))
    t_upload = time.perf_counter() - t1
    marks = {}
    cb_t = C.CFUNCTYPE(None, C.c_int, C.c_int, C.c_void_p)
    cb = cb_t(lambda stage, begin, _u: marks.__setitem__((stage, begin), time.perf_counter()))
    rb = (C.c_uint32 * n)(*bounds[:n])

    def step():
        runs = [host.Run(ds, prm, tables[r], None) for r in range(n)]
        hs = (C.c_void_p * n)(*[r._h for r in runs])
        ts = time.perf_counter()
        if H.hxh_runs_all_sharded(hs, n, rb, cb, None) != 0:
            raise SystemExit("hxh_runs_all_sharded: " + H.hxh_last_error().decode())
        return runs, marks[(3, 0)] - ts, {k: marks[(k, 0)] - marks[(k, 1)] for k in range(5)}

    for _ in range(args.warmup):
        for r in step()[0]:
            r.close()
    for c in ctxs:
        L.hx_timing_reset(c)
    total, last, stages = 0.0, None, None
    for _ in range(args.steps):
        if last:
            for r in last:
                r.close()
        last, dt, stages = step()
        total += dt
        log(f"[group of {n}] step {dt:.3f} s  stages {stages}")
    by, ms = C.c_uint64(), C.c_double()
    L.hx_group_exchange_stats(g, C.byref(by), C.byref(ms))
    fasta = last[0].assembly_fasta()
    sha = hashlib.sha256(fasta.encode()).hexdigest()
    st_all = [r.cns_stats() for r in last]
    cells = sum(x["dp_cells"] for x in st_all)
    alg_bytes = sum((x["seq_bases"] + 3) // 4 for x in st_all) + sum(sum(len(q) for q in r.cns_out()) for r in last)   # SURVEY 8d, all ranks
    poa_ms_rank = []
    for c in ctxs:
        tm, nl = (C.c_double * 4)(), (C.c_uint64 * 4)()
        L.hx_timing_get(c, C.byref(tm), C.byref(nl))
        poa_ms_rank.append(tm[3] / max(1, nl[3]))
    poa_ms = max(poa_ms_rank)   # (the slowest rank's POA launch group per step)
    n_edges = last[0].n_edges_total
    for r in last:
        r.close()
    L.hx_group_destroy(g)
    if not solo_first:
        solo_sha, t_solo, n1 = solo_reference(0)   # (its first whole pass is the warm-up)
        n1["what"] = "the plain single-context path (bench.py's N = 1 step) over the same data on device 0, timed here after the group's steps"
    same = solo_sha == sha
    value = ds.total_read_bases * args.steps / total
    achieved = alg_bytes / (poa_ms / 1e3) / 1e9 if poa_ms > 0 else 0.0
    gcups = cells / (poa_ms / 1e3) / 1e9 if poa_ms > 0 else 0.0
    valu_bound = VALU_PEAK_LANE_OPS / VALU_OPS_PER_CELL_MODEL / 1e9
    line = {"metric": "long-read bases/sec through backbone+consensus; GFA match + FASTA %identity", "value": value, "unit": "long-read bases/s",
            "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": total / args.steps * 1e3, "higher_is_better": True,
            "scaling": "single GPU" if n == 1 else "fixed-size (configs[3] as named)" if as_named else "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"{wl['name']}: {glen} bp genome in {chroms} chromosome(s), {wl['model']}-like 25x long reads + PAF vs short-read contigs (BASELINE.json configs[{wl['config']}]"
                                   + (", read-sharded over %d GPUs as BASELINE names it)" % n if as_named else " x%d, read-sharded)" % n),
                       "launch": f"in-process group: {n} ranks as threads of this process (hx_group_create / hx_edge_merge / hxh_runs_all_sharded = haslr_assemble --gpus {n}), transport {transport}",
                       "reads": ds.reads.n, "long_read_bases": ds.total_read_bases, "paf_records": ds.hits.n, "edges": int(n_edges),
                       "parallelism": f"reads+edges sharded x{n}, 1 all-gather of edge records ({transport}), results through process memory",
                       "rccl_ranks": list(rr), "edge_record_exchange_bytes": by.value, "edge_record_exchange_ms": ms.value,
                       "n1_same_data_value": n1["value"], "value_over_n1_same_data": value / n1["value"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS * n, "unit": "GB/s", "frac": achieved / (HBM_PEAK_GBS * n), "traffic": None,
                         "kernel": "k_poa (launch group of the slowest rank)", "kernel_ms_per_launch": poa_ms, "kernel_ms_per_launch_by_rank": poa_ms_rank,
                         "algorithmic_bytes_per_launch": alg_bytes, "gcups": gcups, "dp_cells_per_launch": cells, "valu_bound_gcups": valu_bound * n, "frac_of_valu_bound": gcups / (valu_bound * n),
                         "note": "all ranks' algorithmic bytes / cells over the slowest rank's POA launch group; peak = N x one GPU's"
                                 + ("; the ranks SHARE one device in this rehearsal (transport host): their kernels interleave, the figure says nothing about N devices" if transport == "host" and n > 1 else "")},
            "stage_s": stages, "dp_cells_per_step": cells, "n1_same_data": n1,
            "assembly": {"sha256": sha, "contigs": fasta.count(">"), "matches_single_gpu": same, "single_gpu_pass_s": t_solo},
            "ingest": {"seconds": t_parse, "upload_seconds_all_ranks": t_upload}}
    if not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_sample(ds, share=max(8, 4 * n))
        except Exception as e:  # noqa: BLE001
            line["cpu_baseline"] = {"error": str(e)}
    print(json.dumps(line), flush=True)
    if not same:
        raise SystemExit("in-process group: the sharded assembly differs from the single-GPU pass")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default=None,
                    help="default: yeast (BASELINE configs[2]; x N for N = 2, 8), fly (configs[3] as BASELINE names it: 140 Mb on 4 GPUs) at N = 4")
    ap.add_argument("--genome-len", type=int, default=0, help="override the per-run genome length (testing)")
    ap.add_argument("--poa-block", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs1", action="store_true", help="skip the extra E. coli-size (configs[1]) measurement")
    ap.add_argument("--no-one-shot", action="store_true", help="skip the one-shot figures (haslr_assemble end to end in a process of its own, the cold first pass of a fresh context)")
    ap.add_argument("--no-configs3", action="store_true", help="skip the extra D. melanogaster-size (configs[3], 140 Mb on this one GPU: the many-edge regime) measurement")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from haslr_amd import hip, host

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # HASLR_BENCH_FORCE_DIST=1 (testing): the N>1 code path - process group, record all-gather, results exchange - with whatever world size the launcher
    # gave, also 1: a one-GPU box then executes the RCCL collectives of the multi-GPU path for real
    dist_on = world > 1 or os.environ.get("HASLR_BENCH_FORCE_DIST") == "1"
    if world == 1 and (args.gpus > 1 or os.environ.get("HASLR_BENCH_FORCE_GROUP") == "1"):
        return run_group(args)                     # no launcher: the ranks live inside this process, like haslr_assemble --gpus N (HASLR_BENCH_FORCE_GROUP=1: also for one rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    # HASLR_DIST_BACKEND=gloo (testing): the N>1 path on a box with fewer GPUs than ranks - ranks share devices, collectives go through host memory
    backend_name = os.environ.get("HASLR_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend_name == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    comm_device = torch.device("cpu") if backend_name == "gloo" else device
    torch.zeros(1, device="cuda")   # initialise torch's HIP context (streams, queues) now, not inside the timed region
    torch.cuda.synchronize()
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend_name == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    args.workload, glen, chroms, as_named = multi_gpu_workload(world, args.workload, args.genome_len)
    wl = WORKLOADS[args.workload]
    if rank == 0:
        make_dataset(wl, glen, "gpu", chroms)
    if dist_on:
        dist.barrier()
    pre = make_dataset(wl, glen, "gpu", chroms)
    t0 = time.perf_counter()
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")   # multi-threaded ingest, automatic thread count
    t_parse = time.perf_counter() - t0
    in_bytes = sum(os.path.getsize(pre + x) for x in (".contigs.fa", ".reads.fa", ".paf"))
    prm = ds.params()
    one_shot = {}
    if world == 1 and not dist_on and not args.no_one_shot:
        # the drop-in binary, once, in a process of its own - before this process holds a context (the device is the binary's alone, as in real use)
        one_shot["cli"] = cli_e2e(pre, args.workload)
        log(f"[rank {rank}] haslr_assemble end to end: {one_shot['cli']}")
    ctx = hip.HipContext(dev_index)
    if args.poa_block:
        ctx.set_poa_block(args.poa_block)
    up = upload_and_reserve(ctx, ds)   # inputs resident in HBM before the timed region
    t_upload = up["upload_s"]
    log(f"[rank {rank}] dataset {pre}: {ds.reads.n} reads, {ds.total_read_bases} bases, {ds.hits.n} PAF records; parse {t_parse:.2f} s ({in_bytes / 1e6 / t_parse:.0f} MB/s), upload {t_upload:.2f} s, workspace reserved beside it in {up['workspace_reserve_s']:.2f} s")
    if world == 1 and not dist_on and not args.no_one_shot:
        one_shot.update(cold_pass(ctx, ds, prm, ctx.backend()))
        one_shot["start_up"] = up
        log(f"[rank {rank}] cold first pass {one_shot['cold_first_pass_ms']:.1f} ms (consensus {one_shot['cold_consensus_ms']:.1f} ms; host parts {one_shot['cold_poa_host_ms']}; arena {one_shot['workspace_arena']})")

    lr_begin, gathered = 0, [0]
    if dist_on:
        from haslr_amd import distributed as hd
        b = hd.shard_bounds(ds.read_hit_off, ds.reads.n, world)
        ctx.set_read_shard(b[rank], b[rank + 1])
        lr_begin = b[rank]
        backend = hd.ShardedBackend(ctx.backend(), hd.HipRecords(ctx, prm, comm_device))
        table = backend.table

        def gather(run):
            gathered[0] = hd.sharded_stages(run, comm_device, backend=backend)
    else:
        table, gather = ctx.backend(), None

    def sync():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    dt, last = measure(ctx, ds, prm, table, args.steps, args.warmup, world, rank, sync, gather, lr_begin, sharded=dist_on)
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-launch figures of the dominant kernel (K6 POA), hipEvent-timed inside the library on its stream
    tim = ctx.timing()
    st = last.cns_stats()
    cns = last.cns_out()
    poa_ms = tim["poa"]["ms"] / max(1, tim["poa"]["launches"])
    alg_bytes = (st["seq_bases"] + 3) // 4 + sum(len(c) for c in cns)   # 2-bit gap bases read once + consensus written once
    stats = torch.tensor([st["dp_cells"], st["seq_bases"], alg_bytes, last.n_edges], dtype=torch.float64, device=comm_device)
    tms = torch.tensor([poa_ms], dtype=torch.float64, device=comm_device)
    if dist_on:
        dist.all_reduce(stats)
        dist.all_reduce(tms, op=dist.ReduceOp.MAX)
    cells, seq_bases, alg_bytes, n_edges = [float(x) for x in stats.tolist()]
    poa_ms = float(tms.item())
    achieved = alg_bytes / (poa_ms / 1e3) / 1e9 if poa_ms > 0 else 0.0
    phase = ctx.poa_phase_cycles()
    stage_ms = {k: v * 1e3 for k, v in last.timings().items()}
    poa_host = ctx.poa_host_times()   # of the last step's consensus call
    kernel_ms = {k: v["ms"] / max(1, v["launches"]) for k, v in tim.items()}

    # ---- the assembly (outside the timed region): every rank stitches; N>1: identical on all ranks and equal to a single-GPU pass
    last.assemble()
    fasta = last.assembly_fasta()
    sha = hashlib.sha256(fasta.encode()).hexdigest()
    assembly = {"sha256": sha, "contigs": fasta.count(">"), "bases": sum(len(x) for x in fasta.split("\n") if x and x[0] != ">")}
    if dist_on:
        h = torch.frombuffer(bytearray(bytes.fromhex(sha)), dtype=torch.uint8).to(comm_device)
        hs = [torch.zeros(32, dtype=torch.uint8, device=comm_device) for _ in range(world)]
        dist.all_gather(hs, h)
        assembly["same_on_all_ranks"] = all(bool(torch.equal(hs[0], x)) for x in hs)
        assembly["results_gathered_bytes"] = gathered[0]
        assembly["edge_record_exchange_bytes"] = backend.exchange_bytes
        assembly["edge_record_exchange_ms"] = backend.exchange_ms      # wall time of the record all-gather of the last step on this rank (over xGMI with RCCL; SURVEY 8e expects 10-15 ms at CHM1 scale)
        if backend_name == "gloo" and dist.get_world_size() > 1:
            # rehearsal mode (all ranks on one device): the single-GPU pass needs the memory the other ranks' workspaces hold
            last.close()
            ctx.close()
            torch.cuda.empty_cache()
            dist.barrier()
            if rank == 0:
                ctx = hip.HipContext(dev_index)
                ctx.upload(ds)
        if rank == 0:
            last.close()
            ctx.set_read_shard(0, ds.reads.n)
            t0 = time.perf_counter()
            solo = host.Run(ds, prm, ctx.backend(), None)
            solo.all()
            assembly["single_gpu_pass_s"] = time.perf_counter() - t0
            assembly["matches_single_gpu"] = hashlib.sha256(solo.assembly_fasta().encode()).hexdigest() == sha
            last = solo
        dist.barrier()
        if not assembly["same_on_all_ranks"] or (rank == 0 and not assembly["matches_single_gpu"]):
            raise SystemExit(f"[rank {rank}] multi-GPU assembly differs: {assembly}")

    # HBM traffic of the POA launch group: measured offline with rocprofv3 PMC passes for exactly this workload (profiles/*_traffic.json)
    traffic, traffic_src = None, None
    if world == 1 and not args.genome_len:
        for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
            try:
                t = json.load(open(cand))
                if t.get("bench_workload", "ecoli") == args.workload:
                    traffic, traffic_src = t["hbm_bytes_raw"], os.path.relpath(cand, ROOT)
                    break
            except Exception:  # noqa: BLE001
                pass
    if rank == 0:
        value = ds.total_read_bases * args.steps / dt
        gcups = cells / (poa_ms / 1e3) / 1e9 if poa_ms > 0 else 0.0
        valu_bound = VALU_PEAK_LANE_OPS / VALU_OPS_PER_CELL_MODEL / 1e9 * world   # (N > 1: all ranks' cells over the slowest rank's launch group, against N GPUs)
        # the longest edge's serial chain of {decode, DP, traceback, graph update, order update, CSR rebuild}: lane-0 cycle counters of the kernel
        critical_ms = sum(phase["slowest_edge"].values()) / SHADER_CLOCK_HZ * 1e3
        # SQ counters of the POA launch group, collected offline for exactly this workload (profiles/*_sq_counters.json: separate --pmc passes)
        sq, sq_src = {}, None
        if world == 1 and not args.genome_len:
            for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_counters.json")), reverse=True):
                try:
                    t = json.load(open(cand))
                    if t.get("bench_workload", "yeast") == args.workload:
                        sq, sq_src = t["counters_summed_over_k_poa_dispatches"], os.path.relpath(cand, ROOT)
                        break
                except Exception:  # noqa: BLE001
                    pass
        sq_fig = {}
        if sq.get("SQ_INSTS_VALU") and cells:
            # profiled cells = this line's cells (same workload, same data set); a wave64 VALU instruction = 64 lane-ops, 2 cycles of a SIMD-32
            prof_ms = sq.get("profiled_kernel_ms", poa_ms)
            sq_fig = {"valu_lane_ops_per_cell": sq["SQ_INSTS_VALU"] * 64 / cells,
                      "valu_issue_util": sq["SQ_INSTS_VALU"] * 2 / (256 * 4 * SHADER_CLOCK_HZ * prof_ms / 1e3),
                      "wait_share": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"] if sq.get("SQ_WAVE_CYCLES") else None,
                      "sq_counters_source": sq_src}
        gfa_gpu = None
        if world == 1:   # the metric's "GFA match": the six backbone.0x.*.gfa snapshots of a GPU pass (untimed) against the CPU leg's, byte for byte
            work_dir = os.path.join(os.environ.get("HASLR_BENCH_DIR", "/tmp/haslr_bench"), "gfa_%s" % args.workload)
            r2 = host.Run(ds, prm, ctx.backend(), os.path.join(work_dir, "gpu_gfa"))
            r2.chain(); r2.graph(); r2.close()
            gfa_gpu = gfa_digest(os.path.join(work_dir, "gpu_gfa"))
        line = {
            "metric": "long-read bases/sec through backbone+consensus; GFA match + FASTA %identity",
            "value": value, "unit": "long-read bases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "single GPU" if world == 1 else "fixed-size (configs[3] as named)" if as_named else "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"{wl['name']}: {glen} bp genome, {wl['model']}-like 25x long reads + PAF vs short-read contigs "
                                   f"(BASELINE.json configs[{wl['config']}]{(', read-sharded over %d GPUs as BASELINE names it' % world) if as_named else ((' x%d = %d chromosomes, read-sharded' % (world, chroms)) + ('; the shape of configs[4] at %.2f x its size' % (glen / 3.1e9) if world == 8 else '') if world > 1 else '')})",
                       "reads": ds.reads.n, "long_read_bases": ds.total_read_bases, "paf_records": ds.hits.n, "edges": int(n_edges),
                       "poa_block_threads": args.poa_block or "auto", "parallelism": f"reads+edges sharded x{world}, 1 all-gather of edge records + 1 of results" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": achieved / (HBM_PEAK_GBS * world),
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "k_poa (launch group: one kernel per lane-count class, concurrent)", "kernel_ms_per_launch": poa_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "gcups": gcups, "dp_cells_per_launch": cells, "poa_workspace_bytes": ctx.poa_workspace_bytes(), "valu_bound_gcups": valu_bound, "frac_of_valu_bound": gcups / valu_bound,
                         "valu_peak_lane_ops_per_s": VALU_PEAK_LANE_OPS, "valu_ops_per_cell_model": VALU_OPS_PER_CELL_MODEL, **sq_fig,
                         "critical_path_ms": critical_ms, "critical_path_share_of_launch": critical_ms / poa_ms if poa_ms > 0 else None, "pruning": ctx.poa_prune_stats(),
                         "note": "POA is an O(L^2) integer DP over O(L) bytes: HBM fraction is low by construction (SURVEY.md 8d); GCUPS against the VALU issue bound "
                                 "(256 CU x 4 SIMD-32 x 2.4 GHz lane-ops/s / 10 lane-ops per cell) is the figure of merit; valu_lane_ops_per_cell / valu_issue_util / wait_share are "
                                 "what the SQ counters of the committed profile say the kernel really issues; critical_path_ms = the slowest edge's serial chain (its lane-0 cycle counters / 2.4 GHz)"},
            "stage_ms": stage_ms, "kernel_ms": kernel_ms, "consensus_host_ms": {**poa_host, "stage_minus_launch_group_ms": stage_ms.get("consensus", 0) - poa_ms}, "poa_phase_cycles": {"edges": phase["edges"], "slowest_edge": phase["slowest_edge"]}, "assembly": assembly, "gfa": gfa_gpu,
            # outside the timed region (the metric starts with parsed, resident inputs): text ingest and the PCIe upload
            "ingest": {"seconds": t_parse, "input_mb": in_bytes / 1e6, "mb_per_s": in_bytes / 1e6 / t_parse, "threads": os.environ.get("HASLR_IO_THREADS", "auto (<= 16)"), "upload_seconds": t_upload},
        }
        if world > 1:
            line["config"].update({"edge_record_exchange_ms": assembly.get("edge_record_exchange_ms"), "edge_record_exchange_bytes": assembly.get("edge_record_exchange_bytes"),
                                   "rccl_ranks": dist.get_world_size() if backend_name == "nccl" else 0})
            if not args.no_cpu_baseline:
                try:
                    line["cpu_baseline"] = cpu_sample(ds, share=max(8, 4 * world))
                except Exception as e:  # noqa: BLE001
                    line["cpu_baseline"] = {"error": str(e)}
        if one_shot:
            line["one_shot"] = one_shot
            ms_step = line["ms_per_step"]
            line["config"].update({"cold_first_pass_ms": one_shot.get("cold_first_pass_ms"), "cold_consensus_ms": one_shot.get("cold_consensus_ms"),
                                   "cold_over_steady": (one_shot["cold_first_pass_ms"] / ms_step) if one_shot.get("cold_first_pass_ms") else None,
                                   "cli_e2e_s": one_shot.get("cli", {}).get("cli_e2e_s"), "cli_stages_s": one_shot.get("cli", {}).get("stages_s"),
                                   "cli_assembly_equals_in_process": one_shot.get("cli", {}).get("asm_final_fa_sha256") == sha})
        if not args.no_cpu_baseline and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(ds, cns, gfa_gpu, work_dir)
                line["gfa_equals_cpu_baseline"] = line["cpu_baseline"]["gfa_equals_gpu"]   # six files, byte for byte (the CPU leg = the oracle, pinned to the compiled reference for the GFAs)
            except Exception as e:  # noqa: BLE001
                line["cpu_baseline"] = {"error": str(e)}
        if not args.no_configs1 and world == 1 and args.workload != "ecoli" and not args.genome_len:
            # BASELINE.json configs[1] (the correctness-gate configuration) on the same GPU: short extra measurement
            try:
                last.close(); ds.close()
                w1 = WORKLOADS["ecoli"]
                pre1 = make_dataset(w1, w1["genome"], "gpu")
                ds1 = host.Dataset(pre1 + ".contigs.fa", pre1 + ".reads.fa", pre1 + ".paf")
                ctx.upload(ds1)
                dt1, run1 = measure(ctx, ds1, ds1.params(), ctx.backend(), 3, 1, 1, 0, sync, None, 0)
                tm1 = ctx.timing()
                c1 = run1.cns_stats()["dp_cells"]
                p1 = tm1["poa"]["ms"] / max(1, tm1["poa"]["launches"])
                line["configs1"] = {"workload": f"{w1['name']}: {w1['genome']} bp genome, pacbio-like 25x (BASELINE.json configs[1])", "value": ds1.total_read_bases * 3 / dt1,
                                    "ms_per_step": dt1 / 3 * 1e3, "steps": 3, "warmup": 1, "edges": run1.n_edges, "kernel_ms_per_launch": p1, "gcups": c1 / (p1 / 1e3) / 1e9,
                                    "long_read_bases": ds1.total_read_bases}
                line["config"]["configs1_ms_per_step"] = line["configs1"]["ms_per_step"]   # (inside `config`: the driver's record keeps that object)
                run1.close(); ds1.close()
                last = None
            except Exception as e:  # noqa: BLE001
                line["configs1"] = {"error": str(e)}
        if not args.no_configs3 and world == 1 and args.workload != "fly" and not args.genome_len:
            # BASELINE.json configs[3]'s data set (140 Mb, PacBio-like 25x) on this ONE GPU: the many-edge regime (13 000 edges, every CU busy) - the regime the
            # CHM1 target lives in, where the step is throughput and not one edge's serial chain. ~60 s to simulate, ~3 s to parse, 3 passes.
            try:
                if last is not None:
                    last.close(); last = None
                try:
                    ds.close()
                except Exception:  # noqa: BLE001
                    pass
                w3 = WORKLOADS["fly"]
                t0 = time.perf_counter()
                pre3 = make_dataset(w3, w3["genome"], "gpu")
                t_sim = time.perf_counter() - t0
                # a FRESH context for this leg (the one-shot figures need one), and before it exists the binary itself, once, on the same files
                ctx.close()
                torch.cuda.empty_cache()
                one3 = {}
                if not args.no_one_shot:
                    one3["cli"] = cli_e2e(pre3, "fly")
                    log(f"[configs3] haslr_assemble end to end: {one3['cli']}")
                t0 = time.perf_counter()
                ds3 = host.Dataset(pre3 + ".contigs.fa", pre3 + ".reads.fa", pre3 + ".paf")
                t_parse3 = time.perf_counter() - t0
                ctx = hip.HipContext(dev_index)
                up3 = upload_and_reserve(ctx, ds3)
                if not args.no_one_shot:
                    one3.update(cold_pass(ctx, ds3, ds3.params(), ctx.backend()))
                    one3["start_up"] = up3
                    log(f"[configs3] cold first pass {one3['cold_first_pass_ms']:.1f} ms (consensus {one3['cold_consensus_ms']:.1f} ms; host parts {one3['cold_poa_host_ms']}; arena {one3['workspace_arena']})")
                dt3, run3 = measure(ctx, ds3, ds3.params(), ctx.backend(), 2, 1, 1, 0, sync, None, 0)
                host3 = ctx.poa_host_times()
                tm3 = ctx.timing()
                st3 = run3.cns_stats()
                p3 = tm3["poa"]["ms"] / max(1, tm3["poa"]["launches"])
                cns3 = run3.cns_out()
                h3 = hashlib.sha256()
                for c in cns3:
                    h3.update(c if isinstance(c, bytes) else str(c).encode()); h3.update(b"\n")
                ph3 = ctx.poa_phase_cycles()
                sq3 = {}
                for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_counters.json")), reverse=True):
                    try:
                        t = json.load(open(cand))
                        if t.get("bench_workload") == "fly":
                            q = t["counters_summed_over_k_poa_dispatches"]
                            sq3 = {"valu_lane_ops_per_cell": q["SQ_INSTS_VALU"] * 64 / st3["dp_cells"], "sq_counters_source": os.path.relpath(cand, ROOT),
                                   "valu_issue_util": q["SQ_INSTS_VALU"] * 2 / (256 * 4 * SHADER_CLOCK_HZ * q.get("profiled_kernel_ms", p3) / 1e3)}
                            break
                    except Exception:  # noqa: BLE001
                        pass
                line["configs3"] = {"workload": f"{w3['name']}: {w3['genome']} bp genome, pacbio-like 25x (BASELINE.json configs[3]'s data set, here on ONE GPU)",
                                    "value": ds3.total_read_bases * 2 / dt3, "ms_per_step": dt3 / 2 * 1e3, "steps": 2, "warmup": 1, "edges": run3.n_edges,
                                    "kernel_ms_per_launch": p3, "gcups": st3["dp_cells"] / (p3 / 1e3) / 1e9, "dp_cells_per_launch": st3["dp_cells"],
                                    "long_read_bases": ds3.total_read_bases, "poa_workspace_bytes": ctx.poa_workspace_bytes(), "consensus_sha256": h3.hexdigest(),
                                    "critical_path_ms": sum(ph3["slowest_edge"].values()) / SHADER_CLOCK_HZ * 1e3, "stage_ms": {k: v * 1e3 for k, v in run3.timings().items()},
                                    "simulate_s": t_sim, "parse_s": t_parse3, "pruning": ctx.poa_prune_stats(), **sq3,
                                    "consensus_host_ms": {**host3, "stage_minus_launch_group_ms": run3.timings()["consensus"] * 1e3 - p3}, "one_shot": one3}
                # the many-edge figures inside `config` and `roofline`, the objects the driver's record keeps (BENCH_rNN.json.parsed)
                if one3:
                    line["config"].update({"configs3_cold_first_pass_ms": one3.get("cold_first_pass_ms"), "configs3_cold_consensus_ms": one3.get("cold_consensus_ms"),
                                           "configs3_cold_over_steady": one3["cold_first_pass_ms"] / line["configs3"]["ms_per_step"] if one3.get("cold_first_pass_ms") else None,
                                           "configs3_cli_e2e_s": one3.get("cli", {}).get("cli_e2e_s"), "configs3_cli_stages_s": one3.get("cli", {}).get("stages_s"),
                                           "configs3_consensus_stage_minus_launch_group_ms": line["configs3"]["consensus_host_ms"]["stage_minus_launch_group_ms"]})
                line["config"].update({"configs3_ms_per_step": line["configs3"]["ms_per_step"], "configs3_gcups": line["configs3"]["gcups"], "configs3_edges": run3.n_edges,
                                       "configs3_value": line["configs3"]["value"], "configs3_consensus_sha256": h3.hexdigest()})
                pr3 = line["configs3"]["pruning"]
                line["roofline"].update({"configs3_kernel_ms_per_launch": p3, "configs3_gcups": line["configs3"]["gcups"], "configs3_frac_of_valu_bound": line["configs3"]["gcups"] / valu_bound,
                                         "configs3_wave_rows_skipped": (pr3["wave_rows_skipped"] / pr3["wave_rows"]) if pr3["wave_rows"] else None,
                                         "configs3_valu_lane_ops_per_cell": sq3.get("valu_lane_ops_per_cell")})
                run3.close(); ds3.close()
            except Exception as e:  # noqa: BLE001
                line["configs3"] = {"error": str(e)}
        print(json.dumps(line), flush=True)
    if last is not None:
        last.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
