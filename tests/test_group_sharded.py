"""The multi-GPU mode of the drop-in binary (haslr_assemble --gpus N): N ranks INSIDE one process, one host thread per rank
(libhaslr_host.so hxh_runs_all_sharded over the ranks' backend tables; on the device side include/haslr_hip.h hx_group_* / hx_edge_merge,
one RCCL all-gather of the edge-support records). Replaces the reference's worker threads behind asm_calc_edge_coordinates_MT /
asm_cal_cns_seq_MT (Assemble.cpp:453-477, :580-605; main.cpp:203-208).
  CPU: the host orchestration over oracle backends - every output file equal to a single rank's; an error on one rank ends all ranks.
  GPU: the binary with --gpus 2 on one device (host-staged exchange), and --gpus 1 through the group path over RCCL (ncclCommInitAll +
       ncclAllGather really execute), both byte-equal to the plain single-GPU run."""
import ctypes as C
import os
import subprocess

import pytest

import orclib
import util
from haslr_amd import ctypes_defs as T
from haslr_amd import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "haslr_amd", "bin", "haslr_assemble")


def _sharded_oracle_pass(ds, n, out, threads=2):
    b = host.shard_bounds(ds, n)
    backs, runs = [], []
    for r in range(n):
        ob = orclib.OracleBackend(ds, threads)
        ob.set_read_shard(b[r], b[r + 1])      # the rank chains its own reads only; the oracle's edge_support still returns the merged multiset
        backs.append(ob)
        runs.append(host.Run(ds, ds.params(), ob.table, out if r == 0 else None))
    host.runs_all_sharded(runs, b)
    return b, backs, runs


@pytest.mark.parametrize("n", [2, 3, 5])
def test_in_process_ranks_write_the_single_rank_outputs(n, sim, tmp_path):
    pre = sim("--genome-len", "150000", "--seed", "61", "--variant-per-mb", "30", "--cov", "14")
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
    one = orclib.OracleBackend(ds, 4)
    lone = host.Run(ds, ds.params(), one.table, str(tmp_path / "one"))
    lone.all()
    b, backs, runs = _sharded_oracle_pass(ds, n, str(tmp_path / "many"))
    assert b[0] == 0 and b[-1] == ds.reads.n and sorted(b) == b
    assert util.compare_dirs(str(tmp_path / "one"), str(tmp_path / "many")) == []          # GFAs, stats, logs, compact_uniq.txt, asm.final.*
    assert runs[0].assembly_fasta() == lone.assembly_fasta() and len(lone.assembly_fasta()) > 1000
    shares = [r.n_edges for r in runs]
    assert sum(shares) == runs[0].n_edges_total and runs[0].results_missing == 0            # the shares partition the work queue
    if n <= 3:
        assert all(0 < s < runs[0].n_edges_total for s in shares), shares
    for r in runs + [lone]:      # (runs before their backends: a run hands its stage outputs back to the backend that made them)
        r.close()
    for ob in backs + [one]:
        ob.close()
    ds.close()


def test_error_on_one_rank_ends_all_ranks(sim, tmp_path):
    """a failing stage on one rank: every rank stops after that stage (nobody is left waiting), the error names the rank"""
    pre = sim("--genome-len", "100000", "--seed", "62", "--variant-per-mb", "30", "--cov", "12")
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
    b = host.shard_bounds(ds, 3)
    backs = [orclib.OracleBackend(ds, 1) for _ in range(3)]
    for r, ob in enumerate(backs):
        ob.set_read_shard(b[r], b[r + 1])
    calls = []
    proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(T.CoordsOut))

    def broken_coords(_ctx, _n, _sel, _out):
        calls.append(1)
        return -1

    cb = proto(broken_coords)
    table = T.Backend()
    C.memmove(C.byref(table), C.byref(backs[1].table), C.sizeof(T.Backend))
    table.edge_coords = C.cast(cb, C.c_void_p).value
    runs = [host.Run(ds, ds.params(), table if r == 1 else backs[r].table, None) for r in range(3)]
    with pytest.raises(host.HostError, match="rank 1"):
        host.runs_all_sharded(runs, b)
    assert calls == [1]
    for r in runs:
        r.close()
    for ob in backs:
        ob.close()


def _cli(args, env=None, timeout=600):
    return subprocess.run([BIN] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, **(env or {})), timeout=timeout)


@pytest.mark.gpu
def test_binary_with_two_ranks_on_one_gpu_and_one_rank_over_rccl(sim, built, tmp_path):
    pre = sim("--genome-len", "400000", "--seed", "63", "--variant-per-mb", "20", "--cov", "20")
    base = ["-t", "8", "-c", pre + ".contigs.fa", "-l", pre + ".reads.fa", "-m", pre + ".paf"]
    one = _cli(base + ["-d", str(tmp_path / "one")])
    assert one.returncode == 0, one.stderr[-2000:]
    # two ranks, both on device 0, records exchanged through host memory: the sharding / merge / results exchange / stitching logic
    two = _cli(base + ["-d", str(tmp_path / "two"), "--gpus", "2"], {"HASLR_GROUP_TRANSPORT": "host"})
    assert two.returncode == 0, two.stderr[-2000:]
    assert "2 GPU ranks in this process, edge-record exchange over host" in two.stderr and "exchanged" in two.stderr
    # one rank through the group path over RCCL: ncclCommInitAll and ncclAllGather execute on the hardware
    rc1 = _cli(base + ["-d", str(tmp_path / "rccl1"), "--gpus", "1"], {"HASLR_FORCE_GROUP": "1"})
    assert rc1.returncode == 0, rc1.stderr[-2000:]
    assert "edge-record exchange over rccl" in rc1.stderr
    for other in ("two", "rccl1"):                                # every output file, index.contig / index.longread included
        assert util.compare_dirs(str(tmp_path / "one"), str(tmp_path / other)) == [], other
        assert open(tmp_path / "one" / "index.longread", "rb").read() == open(tmp_path / other / "index.longread", "rb").read()
    assert os.path.getsize(tmp_path / "one" / "asm.final.fa") > 100000
    # more ranks than devices over RCCL is refused, loudly
    bad = _cli(base + ["-d", str(tmp_path / "bad"), "--gpus", "64"], {"HASLR_GROUP_TRANSPORT": "rccl"})
    assert bad.returncode != 0 and "[ERROR]" in bad.stderr


@pytest.mark.gpu
@pytest.mark.skipif(bool(os.environ.get("HASLR_SKIP_HUGE")), reason="HASLR_SKIP_HUGE is set (developer runs)")
def test_binary_with_eight_ranks_on_the_140mb_data_set(sim, built, tmp_path):
    """configs[4]'s world size through the product binary: haslr_assemble --gpus 8 on the 140 Mb PacBio-like data set (BASELINE configs[3]'s), all eight
    ranks on the one device of the box and the record exchange staged through host memory (what one GPU per lease allows) - every rank uploads
    the whole input, chains its eighth of the reads, merges 1/8 of the records, aligns its LPT share of ~13 000 edges inside an eighth of the workspace -
    and every output file equals the --gpus 1 run's."""
    # (the data set of tests/test_gpu_parity.py::test_configs3_full_size_against_oracle under the same key of the session's simulator cache: made once per
    # session, removed here - this test runs after that one)
    pre = sim(*util.CONFIGS3_ARGS)
    try:
        base = ["-t", "16", "-c", pre + ".contigs.fa", "-l", pre + ".reads.fa", "-m", pre + ".paf"]
        one = _cli(base + ["-d", str(tmp_path / "one")])
        assert one.returncode == 0, one.stderr[-2000:]
        eight = _cli(base + ["-d", str(tmp_path / "eight"), "--gpus", "8"], {"HASLR_GROUP_TRANSPORT": "host", "HX_POA_WORKSPACE_GB": "24"}, timeout=1500)
        assert eight.returncode == 0, eight.stderr[-2000:]
        assert "8 GPU ranks in this process" in eight.stderr
        assert util.compare_dirs(str(tmp_path / "one"), str(tmp_path / "eight")) == []
        assert os.path.getsize(tmp_path / "one" / "asm.final.fa") > 100e6
    finally:
        for suffix in (".contigs.fa", ".reads.fa", ".paf", ".genome.fa", ".truth.tsv"):
            try:
                os.remove(pre + suffix)
            except OSError:
                pass


@pytest.mark.gpu
def test_binary_with_more_ranks_than_edges(sim, built, tmp_path):
    """fewer surviving edges than ranks: the idle ranks call hx_edge_coords with no edge and hx_poa_batch with nothing to align, still take part
    in the record exchange and the results exchange, and the outputs are the single-GPU run's"""
    pre = sim("--genome-len", "26000", "--seed", "12", "--cov", "7", "--gap-median", "400")
    base = ["-t", "4", "-c", pre + ".contigs.fa", "-l", pre + ".reads.fa", "-m", pre + ".paf"]
    one = _cli(base + ["-d", str(tmp_path / "one")])
    assert one.returncode == 0, one.stderr[-2000:]
    many = _cli(base + ["-d", str(tmp_path / "many"), "--gpus", "5"], {"HASLR_GROUP_TRANSPORT": "host"})
    assert many.returncode == 0, many.stderr[-2000:]
    assert util.compare_dirs(str(tmp_path / "one"), str(tmp_path / "many")) == []
    assert open(tmp_path / "one" / "index.longread", "rb").read() == open(tmp_path / "many" / "index.longread", "rb").read()
    n_links = sum(1 for ln in open(tmp_path / "one" / "backbone.06.smallbubble.gfa") if ln.startswith("L"))
    assert 0 < n_links // 2 < 5, n_links        # (the point of the case: fewer edges than ranks)


@pytest.mark.gpu
def test_edge_merge_through_the_c_abi_equals_edge_support(sim, built):
    """hx_group_create / hx_edge_merge / hx_group_backend_fill through ctypes: three ranks on one device (host-staged exchange), one Python
    thread per rank; every rank's merged multiset equals the unsharded hx_edge_support"""
    import threading

    import numpy as np
    from haslr_amd import hip
    pre = sim("--genome-len", "150000", "--seed", "21", "--variant-per-mb", "30")
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
    prm = ds.params()
    ctx = hip.HipContext(0)
    ctx.upload(ds)
    ctx.chain_reads(prm)
    whole = ctx.edge_support(prm)
    ctx.close()
    L = hip.lib()
    g = C.c_void_p()
    assert L.hx_group_create(3, None, b"host", C.byref(g)) == 0, L.hx_last_error()
    assert L.hx_group_size(g) == 3 and L.hx_group_transport(g) == b"host"
    b = host.shard_bounds(ds, 3)
    got, errs = [None] * 3, []

    def rank(r):
        try:
            c = L.hx_group_ctx(g, r)
            assert L.hx_upload(c, C.byref(ds.contigs), C.byref(ds.reads), C.byref(ds.hits), ds.read_hit_off) == 0
            assert L.hx_set_read_shard(c, b[r], b[r + 1]) == 0
            ch = T.ChainOut()
            assert L.hx_chain_reads(c, C.byref(prm), C.byref(ch)) == 0
            L.hx_free_chain(c, C.byref(ch))
            e = T.EdgesOut()
            assert L.hx_edge_merge(g, r, C.byref(prm), C.byref(e)) == 0, L.hx_last_error()
            got[r] = T.edges_to_dict(e, sides=True)
            L.hx_free_edges(c, C.byref(e))
        except Exception as ex:  # noqa: BLE001
            errs.append((r, ex))

    th = [threading.Thread(target=rank, args=(r,)) for r in range(3)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for r in range(3):
        for k in whole:
            assert np.array_equal(whole[k], got[r][k]), (r, k)
    L.hx_group_destroy(g)


@pytest.mark.gpu
def test_a_failure_inside_the_collective_ends_the_exchange_with_an_error(sim, built):
    """a rank whose all-gather fails (injected: hx_group_inject_fault) raises the group's abort flag instead of leaving anybody parked on a stream: the
    call returns the error, and the group refuses further exchanges. One rank over real RCCL (the only world size a one-GPU box can run it at); the
    ranks that would be WAITING in a larger group take the same bounded-poll path (hx_api.hip hx_edge_merge)."""
    from haslr_amd import hip
    pre = sim("--genome-len", "150000", "--seed", "21", "--variant-per-mb", "30")
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
    prm = ds.params()
    L = hip.lib()
    g = C.c_void_p()
    assert L.hx_group_create(1, None, b"rccl", C.byref(g)) == 0, L.hx_last_error()
    assert L.hx_group_transport(g) == b"rccl"
    L.hx_group_set_timeout(g, 20.0)
    c = L.hx_group_ctx(g, 0)
    assert L.hx_upload(c, C.byref(ds.contigs), C.byref(ds.reads), C.byref(ds.hits), ds.read_hit_off) == 0
    ch = T.ChainOut()
    assert L.hx_chain_reads(c, C.byref(prm), C.byref(ch)) == 0
    L.hx_free_chain(c, C.byref(ch))
    e = T.EdgesOut()
    assert L.hx_edge_merge(g, 0, C.byref(prm), C.byref(e)) == 0, L.hx_last_error()      # the exchange works ...
    n_ok = e.n_rec
    L.hx_free_edges(c, C.byref(e))
    assert n_ok > 0
    L.hx_group_inject_fault(g, 0)
    assert L.hx_edge_merge(g, 0, C.byref(prm), C.byref(e)) != 0                           # ... an injected failure comes back as an error, at once
    assert b"ncclAllGather" in L.hx_last_error()
    L.hx_group_inject_fault(g, -1)
    assert L.hx_edge_merge(g, 0, C.byref(prm), C.byref(e)) != 0 and b"failed earlier" in L.hx_last_error()   # the group is broken for good
    L.hx_group_destroy(g)
    ds.close()


@pytest.mark.gpu
def test_bench_through_the_group_code_reproduces_the_plain_step(built, tmp_path):
    """`bench.py --gpus 1` sent through the in-process group (HASLR_BENCH_FORCE_GROUP=1, host transport: hx_group_create / hx_edge_merge /
    hxh_runs_all_sharded with one rank) against the plain single-context step over the same data in the same process (`n1_same_data`): the group code adds the
    record export / import and the results hand-over, and must cost nothing beside a step (2 %); the line carries what an N = 1 line carries."""
    import json
    import sys
    env = dict(os.environ, HASLR_BENCH_FORCE_GROUP="1", HASLR_GROUP_TRANSPORT="host", HASLR_BENCH_DIR=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "1", "--no-cpu-baseline"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().split("\n")[-1])
    assert line["n_gpus"] == 1 and line["assembly"]["matches_single_gpu"]
    for key in ("achieved", "peak", "frac", "kernel_ms_per_launch", "algorithmic_bytes_per_launch", "gcups"):
        assert line["roofline"][key] is not None and line["roofline"][key] > 0, key
    assert line["config"]["rccl_ranks"] == [0] and "edge_record_exchange_ms" in line["config"]
    ratio = line["value"] / line["n1_same_data"]["value"]
    assert 0.98 <= ratio <= 1.03, (line["value"], line["n1_same_data"])
