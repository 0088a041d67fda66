"""Full-size configurations of BASELINE.json (configs[2] S. cerevisiae 12 Mb Nanopore 25x, configs[3] D. melanogaster 140 Mb PacBio 25x)
on ONE MI355X: synthetic data of that size (tools/hxsim), the whole path on the GPU, and
  * parity: every per-edge consensus and the final assembly byte-identical to the test oracle's (all host cores), and
  * a size-independent property: the assembled contigs place on the truth genome at high identity (tools/hxident).
One JSON line per configuration (stage timings, sizes, verdicts) on stdout and in gpurun_out/full_size_<name>.json.

    python tools/full_size_check.py yeast        # 12 Mb, nanopore, 25x
    python tools/full_size_check.py fly          # 140 Mb, pacbio, 25x   (minutes; ~25 GB of host memory)
    python tools/full_size_check.py --genome-len 40000000 --model pacbio --cov 25 --name mid
    python tools/full_size_check.py chm1         # BASELINE configs[4]'s WHOLE workload on ONE GPU: 3.1 Gb in 22 chromosomes, PacBio-like 25x, seed 0x4841534c + 4
                                                 # (~190 GB of text - written to /dev/shm when /tmp is too small -, ~82 GB resident in HBM; no full oracle run: the
                                                 # size-independent properties of tests/test_gpu_parity.py::test_configs4_share_full_size_properties, and the CPU oracle
                                                 # on a 1/22 sample of the edges at 64 threads)

This is a measuring / checking tool (like bench.py's cpu_baseline leg it may load the oracle); the product never does."""
import argparse
import json
import os
import subprocess
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PRESETS = {"yeast": dict(genome_len=12_000_000, model="nanopore", cov=25), "fly": dict(genome_len=140_000_000, model="pacbio", cov=25),
           "chm1": dict(genome_len=3_100_000_000, model="pacbio", cov=25, chromosomes=22, seed=0x4841534C + 4, sample=22),
           # configs[4]'s shape at a quarter of its size: what the GPU boxes of this pool hold (their containers are limited to 300 GiB of host memory, tmpfs included:
           # the 3.1 Gb run needs ~190 GB of text and as much again parsed, and the box is lost to the OOM killer three minutes into the simulation). 800 Mb in 8
           # chromosomes: 50 GB of text on /tmp, 4.6e9 CIGAR words (word offsets above 2^32), ~75 000 edges in one consensus call
           "chm1_quarter": dict(genome_len=800_000_000, model="pacbio", cov=25, chromosomes=8, seed=0x4841534C + 4, sample=16),
           "chm1_eighth": dict(genome_len=400_000_000, model="pacbio", cov=25, chromosomes=4, seed=0x4841534C + 4, sample=8),   # (one rank's share of the 8-GPU configuration)
           "chm1_rehearsal": dict(genome_len=60_000_000, model="pacbio", cov=25, chromosomes=4, seed=0x4841534C + 4, sample=4)}   # (the chm1 code path at a size that takes a minute)


def properties(run):
    """the size-independent properties of the edge multiset and the chains (tests/test_gpu_parity.py::test_configs4_share_full_size_properties)"""
    import numpy as np
    e = run.edges_out(sides=False)
    res = {"records_sorted_by_key": bool(np.all(e["key"][1:] >= e["key"][:-1]))}
    ek = e["edge_key"]
    twin = ((ek & np.uint64(0xffffffff)) ^ np.uint64(1)) << np.uint64(32) | ((ek >> np.uint64(32)) ^ np.uint64(1))
    cnt = np.diff(e["edge_off"])
    pos = np.searchsorted(ek, twin)
    res["every_edge_has_a_twin_with_equal_support"] = bool(np.all(pos < ek.size) and np.array_equal(ek[np.minimum(pos, ek.size - 1)], twin) and np.array_equal(cnt[np.minimum(pos, ek.size - 1)], cnt))
    c = run.chain_out()
    a = c["cmp_aln"]
    inner = np.ones(a.size, dtype=bool)
    starts = c["cmp_off"][:-1]
    inner[starts[starts < a.size].astype(np.int64)] = False
    res["chained_hits_never_overlap"] = bool(np.all(c["q_end"][a[:-1]][inner[1:]] <= c["q_start"][a[1:]][inner[1:]]))
    res["edge_records"] = int(e["key"].size); res["edges_before_cleaning"] = int(ek.size); res["alignments_kept"] = int(c["q_end"].size)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("preset", nargs="?", choices=sorted(PRESETS))
    ap.add_argument("--genome-len", type=int)
    ap.add_argument("--model", default="pacbio")
    ap.add_argument("--cov", type=int, default=25)
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--name")
    ap.add_argument("--no-oracle", action="store_true", help="skip the CPU oracle run (parity fields are null)")
    ap.add_argument("--no-sample", action="store_true", help="skip the CPU oracle's sample leg of the chm1 presets")
    ap.add_argument("--no-identity", action="store_true", help="skip the placement of the assembly on the truth genome")
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--cli", action="store_true", help="also run the haslr_assemble binary on the same files (wall time, same assembly)")
    ap.add_argument("--tmp", default="/tmp/full_size")
    ap.add_argument("--reuse", action="store_true", help="keep the data set of an earlier run with the same name")
    a = ap.parse_args()
    cfg = dict(PRESETS[a.preset]) if a.preset else dict(genome_len=a.genome_len, model=a.model, cov=a.cov)
    name = a.name or a.preset or "custom"
    if "seed" in cfg:
        a.seed = cfg["seed"]
    big = cfg["genome_len"] >= 300_000_000 or name == "chm1_rehearsal"
    if big:
        a.no_oracle = True                                        # (the whole data set through the oracle would take most of an hour: a sample below)
        a.passes = max(a.passes, 3)                               # one cold + two steady
        if a.tmp == "/tmp/full_size":                            # ~60 bytes of text per genome base: where there is room for it
            import shutil
            need = 75 * cfg["genome_len"]   # (text + the outputs; /tmp first: a tmpfs counts against the container's memory limit)
            a.tmp = next((d for d in ("/tmp/full_size", "/dev/shm/full_size") if shutil.disk_usage(os.path.dirname(d)).free > need), a.tmp)
    os.makedirs(a.tmp, exist_ok=True)
    pre = os.path.join(a.tmp, name)
    res = {"name": name, "config": cfg, "seed": a.seed}

    def lap(key, t0):
        res[key] = round(time.perf_counter() - t0, 3)
        print(key, res[key], file=sys.stderr, flush=True)

    t0 = time.perf_counter()
    if not (a.reuse and all(os.path.exists(pre + x) for x in (".contigs.fa", ".reads.fa", ".paf"))):
        subprocess.check_call([os.path.join(ROOT, "tools", "hxsim"), "--genome-len", str(cfg["genome_len"]), "--model", cfg["model"], "--cov", str(cfg["cov"]),
                               "--seed", str(a.seed), "--out-prefix", pre] + (["--chromosomes", str(cfg["chromosomes"])] if cfg.get("chromosomes", 1) > 1 else []), stderr=subprocess.DEVNULL)
    lap("simulate_s", t0)
    res["input_bytes"] = {k: os.path.getsize(pre + "." + k) for k in ("contigs.fa", "reads.fa", "paf")}

    from haslr_amd import hip, host
    if os.environ.get("HASLR_DEV_LIBDIR"):                       # (development: A/B against another build of libhaslr_hip.so)
        hip._LIBDIR = os.environ["HASLR_DEV_LIBDIR"]
    t0 = time.perf_counter()
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf", threads=min(64 if big else 32, os.cpu_count() or 1))
    lap("ingest_s", t0)
    res["resident_bytes"] = {"packed_reads": int(ds.reads.off[ds.reads.n]), "cigar_words": 4 * int(ds.hits.cg_off[ds.hits.n])}
    res.update(contigs=int(ds.contigs.n), reads=int(ds.reads.n), hits=int(ds.hits.n), read_bases=int(ds.total_read_bases))
    ctx = hip.HipContext(0)
    t0 = time.perf_counter()
    ctx.upload(ds)
    lap("upload_s", t0)
    if big:   # (what the binary does beside its parse: the consensus workspace's arena ahead of the first call - here after the upload, so that the inputs come first)
        t0 = time.perf_counter()
        ctx.poa_reserve(232 << 30)
        lap("workspace_reserve_s", t0)
    out = os.path.join(a.tmp, name + ".gpu")
    os.makedirs(out, exist_ok=True)
    times = []
    rg = None
    for it in range(a.passes):                                     # first pass sizes the pools, second is the steady state
        if rg is not None:
            rg.close()
        rg = host.Run(ds, ds.params(), ctx.backend(), out if it == a.passes - 1 else None)
        t0 = time.perf_counter()
        rg.chain(); t1 = time.perf_counter()
        rg.graph(); t2 = time.perf_counter()
        rg.coords(); t3 = time.perf_counter()
        rg.consensus(); t4 = time.perf_counter()
        times.append(dict(chain=round(t1 - t0, 3), graph=round(t2 - t1, 3), coords=round(t3 - t2, 3), consensus=round(t4 - t3, 3), hot_path=round(t4 - t0, 3)))
        times[-1]["poa_kernel_ms"] = round(ctx.timing()["poa"]["ms"], 1); ctx.timing_reset()
        times[-1]["poa_host_ms"] = ctx.poa_host_times()
        print("gpu pass", it, times[-1], file=sys.stderr, flush=True)
        if it == 0:
            first_cns, first_edges = rg.cns_out(), rg.n_edges
        elif it == a.passes - 1:
            res["idempotent"] = bool(rg.cns_out() == first_cns and rg.n_edges == first_edges)   # a later pass over the resident data gives the first pass's consensus
        if os.environ.get("HX_DEBUG"):
            ctx.poa_phase_cycles()                            # (prints the kernel's statistics of the last call)
    t0 = time.perf_counter()
    rg.assemble()
    lap("stitch_and_write_s", t0)
    res["gpu_passes"] = times
    res["edges"] = int(rg.n_edges)
    res["bases_per_s"] = round(res["read_bases"] / times[-1]["hot_path"])
    res["dp_cells"] = int(rg.cns_stats()["dp_cells"])
    res["gcups"] = round(res["dp_cells"] / times[-1]["consensus"] / 1e9, 1)
    res["poa_workspace_bytes"] = ctx.poa_workspace_bytes()
    res["poa_memory"] = ctx.poa_memory_stats()
    res["pruning"] = ctx.poa_prune_stats()
    if big:
        t0 = time.perf_counter()
        res["properties"] = properties(rg)
        lap("properties_s", t0)
    asm = rg.assembly_fasta()
    res["assembly"] = {"contigs": asm.count(">"), "bases": sum(len(x) for x in asm.split("\n") if x and x[0] != ">")}

    t0 = time.perf_counter()
    o = "identity nan" if a.no_identity else subprocess.check_output([os.path.join(ROOT, "tools", "hxident"), pre + ".genome.fa", os.path.join(out, "asm.final.fa")]
                                                                     + (["400", str(min(128, os.cpu_count() or 1))] if big else []), text=True)
    lap("identity_s", t0)
    summary = [ln for ln in o.split("\n") if ln.startswith("identity")][-1].split()
    res["identity"] = {summary[i]: float(summary[i + 1]) for i in range(0, len(summary) - 1, 2)}

    if not a.no_oracle:
        import orclib
        threads = os.cpu_count() or 1
        be = orclib.OracleBackend(ds, threads)
        ro = host.Run(ds, ds.params(), be.table, None)
        t0 = time.perf_counter()
        ro.all()
        lap("oracle_s", t0)
        res["oracle_threads"] = threads
        res["parity"] = {"consensus": ro.cns_out() == rg.cns_out(), "assembly": ro.assembly_fasta() == asm}
        ro.close(); be.close()
    else:
        res["parity"] = None
    if cfg.get("sample") and not a.no_sample:   # the CPU oracle on a stated sample: chain + graph over everything, coordinates + consensus over the LPT share 1 / sample of the edges, 64 threads
        import orclib
        k = int(cfg["sample"])
        threads = min(64, os.cpu_count() or 1)
        be = orclib.OracleBackend(ds, threads)
        ro = host.Run(ds, ds.params(), be.table, None)
        ro.set_edge_shard(0, k)
        t0 = time.perf_counter(); ro.chain(); ro.graph(); t_front = time.perf_counter() - t0
        t0 = time.perf_counter(); ro.coords(); ro.consensus(); t_share = time.perf_counter() - t0
        # the same share through the GPU (a run of its own over the resident data): the sample's consensus must be the oracle's
        rs = host.Run(ds, ds.params(), ctx.backend(), None)
        rs.set_edge_shard(0, k)
        rs.chain(); rs.graph(); rs.coords(); rs.consensus()
        est = t_front + k * t_share
        res["cpu_sample"] = {"threads": threads, "share": f"1/{k} of the edges by LPT ({ro.n_edges} of {ro.n_edges_total})", "chain_graph_s": round(t_front, 2), "coords_consensus_share_s": round(t_share, 2),
                             "estimated_whole_s": round(est, 1), "estimated_bases_per_s": round(res["read_bases"] / est), "dp_cells_share": int(ro.cns_stats()["dp_cells"]),
                             "consensus_of_the_share_equals_gpu": bool(ro.cns_out() == rs.cns_out()),
                             "gpu_over_cpu_estimate": round(res["bases_per_s"] / (res["read_bases"] / est), 1)}
        print("cpu sample", res["cpu_sample"], file=sys.stderr, flush=True)
        rs.close(); ro.close(); be.close()
    rg.close(); ds.close()
    ctx.close()                                              # (the binary below needs the device memory this process held)
    if a.cli:
        cli_out = os.path.join(a.tmp, name + ".cli")
        subprocess.call(["rm", "-rf", cli_out])
        t0 = time.perf_counter()
        pr = subprocess.run([os.path.join(ROOT, "haslr_amd", "bin", "haslr_assemble"), "-t", str(min(32, os.cpu_count() or 1)), "-c", pre + ".contigs.fa", "-l", pre + ".reads.fa",
                             "-m", pre + ".paf", "-d", cli_out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        lap("cli_wall_s", t0)
        res["cli"] = {"rc": pr.returncode, "same_assembly": pr.returncode == 0 and open(os.path.join(cli_out, "asm.final.fa")).read() == asm,
                      "stderr_tail": pr.stderr[-1500:]}
    line = json.dumps(res)
    print(line)
    god = os.path.join(ROOT, "gpurun_out")
    os.makedirs(god, exist_ok=True)
    open(os.path.join(god, "full_size_%s.json" % name), "w").write(line + "\n")
    ok = res["parity"] is None or all(res["parity"].values())
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
