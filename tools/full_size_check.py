"""Full-size configurations of BASELINE.json (configs[2] S. cerevisiae 12 Mb Nanopore 25x, configs[3] D. melanogaster 140 Mb PacBio 25x)
on ONE MI355X: synthetic data of that size (tools/hxsim), the whole path on the GPU, and
  * parity: every per-edge consensus and the final assembly byte-identical to the test oracle's (all host cores), and
  * a size-independent property: the assembled contigs place on the truth genome at high identity (tools/hxident).
One JSON line per configuration (stage timings, sizes, verdicts) on stdout and in gpurun_out/full_size_<name>.json.

    python tools/full_size_check.py yeast        # 12 Mb, nanopore, 25x
    python tools/full_size_check.py fly          # 140 Mb, pacbio, 25x   (minutes; ~25 GB of host memory)
    python tools/full_size_check.py --genome-len 40000000 --model pacbio --cov 25 --name mid

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

PRESETS = {"yeast": dict(genome_len=12_000_000, model="nanopore", cov=25), "fly": dict(genome_len=140_000_000, model="pacbio", cov=25)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("preset", nargs="?", choices=sorted(PRESETS))
    ap.add_argument("--genome-len", type=int)
    ap.add_argument("--model", default="pacbio")
    ap.add_argument("--cov", type=int, default=25)
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--name")
    ap.add_argument("--no-oracle", action="store_true", help="skip the CPU oracle run (parity fields are null)")
    ap.add_argument("--no-identity", action="store_true", help="skip the placement of the assembly on the truth genome")
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--cli", action="store_true", help="also run the haslr_assemble binary on the same files (wall time, same assembly)")
    ap.add_argument("--tmp", default="/tmp/full_size")
    ap.add_argument("--reuse", action="store_true", help="keep the data set of an earlier run with the same name")
    a = ap.parse_args()
    cfg = dict(PRESETS[a.preset]) if a.preset else dict(genome_len=a.genome_len, model=a.model, cov=a.cov)
    name = a.name or a.preset or "custom"
    os.makedirs(a.tmp, exist_ok=True)
    pre = os.path.join(a.tmp, name)
    res = {"name": name, "config": cfg, "seed": a.seed}

    def lap(key, t0):
        res[key] = round(time.perf_counter() - t0, 3)
        print(key, res[key], file=sys.stderr, flush=True)

    t0 = time.perf_counter()
    if not (a.reuse and all(os.path.exists(pre + x) for x in (".contigs.fa", ".reads.fa", ".paf"))):
        subprocess.check_call([os.path.join(ROOT, "tools", "hxsim"), "--genome-len", str(cfg["genome_len"]), "--model", cfg["model"], "--cov", str(cfg["cov"]),
                               "--seed", str(a.seed), "--out-prefix", pre], stderr=subprocess.DEVNULL)
    lap("simulate_s", t0)
    res["input_bytes"] = {k: os.path.getsize(pre + "." + k) for k in ("contigs.fa", "reads.fa", "paf")}

    from haslr_amd import hip, host
    if os.environ.get("HASLR_DEV_LIBDIR"):                       # (development: A/B against another build of libhaslr_hip.so)
        hip._LIBDIR = os.environ["HASLR_DEV_LIBDIR"]
    t0 = time.perf_counter()
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf", threads=min(32, os.cpu_count() or 1))
    lap("ingest_s", t0)
    res.update(contigs=int(ds.contigs.n), reads=int(ds.reads.n), hits=int(ds.hits.n), read_bases=int(ds.total_read_bases))
    ctx = hip.HipContext(0)
    t0 = time.perf_counter()
    ctx.upload(ds)
    lap("upload_s", t0)
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
        print("gpu pass", it, times[-1], file=sys.stderr, flush=True)
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
    asm = rg.assembly_fasta()
    res["assembly"] = {"contigs": asm.count(">"), "bases": sum(len(x) for x in asm.split("\n") if x and x[0] != ">")}

    t0 = time.perf_counter()
    o = "identity nan" if a.no_identity else subprocess.check_output([os.path.join(ROOT, "tools", "hxident"), pre + ".genome.fa", os.path.join(out, "asm.final.fa")], text=True)
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
