"""world_size-2 (and one world_size-4) CPU tests (gloo) of the multi-GPU path. The per-shard compute is done by the CPU oracle here (no GPU in
this container); everything around it is the product's: haslr_amd.distributed (read shards, the record all-gather,
ShardedBackend, the results all-gather, run_sharded) and the host pipeline's edge sharding / results export + import /
stitching (libhaslr_host.so).
  * the exchanged record multiset equals the unsharded one
  * a whole sharded pass writes the same asm.final.fa / .ann, the same six GFAs, stats, logs and compact_uniq.txt as one rank
  * stitching refuses to run while other ranks' results are missing
  * more ranks than edges: ranks without work still take part in both all-gathers and stitch the same assembly"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import ctypes as C, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from haslr_amd import host, distributed as hd, ctypes_defs as T
import orclib
pre, out = sys.argv[2], sys.argv[3]
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
prm = ds.params()
L = orclib.lib()
chain = T.ChainOut()
assert L.orc_chain_reads(C.byref(ds.contigs), C.byref(ds.hits), ds.read_hit_off, ds.reads.n, C.byref(prm), C.byref(chain)) == 0
b = hd.shard_bounds(ds.read_hit_off, ds.reads.n, world)
e = T.EdgesOut()
assert L.orc_edge_support(C.byref(ds.contigs), C.byref(ds.hits), C.byref(prm), C.byref(chain), b[rank], b[rank + 1], C.byref(e)) == 0
d = T.edges_to_dict(e, sides=False)
# emission order inside a shard = (read asc, pair asc, fwd before twin); the oracle returns key-sorted (stable), so
# restore emission order by sorting on (lr id, cmp position, twin flag)
lr = d["lr"] & 0x7fffffff
tw = d["lr"] >> 31
pos = np.where(tw == 0, d["cmp_head"], d["cmp_tail"])
order = np.lexsort((tw, pos, lr))
rec = np.zeros((len(order), 4), dtype=np.uint64)
rec[:, 0] = d["key"][order]; rec[:, 1] = d["lr"][order]; rec[:, 2] = d["cmp_head"][order]; rec[:, 3] = d["cmp_tail"][order]
local = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy())
merged, total = hd.allgather_records(local, len(order), 32)
m = merged.numpy().view(np.uint64).reshape(-1, 4)
assert total == m.shape[0]
idx = np.argsort(m[:, 0], kind="stable")
m = m[idx]
if rank == 0:
    np.save(out, m)
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_exchange_reproduces_single_rank_multiset(sim, built, tmp_path):
    import ctypes as C
    import orclib
    from haslr_amd import ctypes_defs as T
    from haslr_amd import host
    pre = sim("--genome-len", "120000", "--seed", "77", "--variant-per-mb", "30", "--cov", "12")
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    out = str(tmp_path / "merged.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", str(w), ROOT, pre, out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    m = np.load(out)
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
    prm = ds.params()
    L = orclib.lib()
    chain, e = T.ChainOut(), T.EdgesOut()
    assert L.orc_chain_reads(C.byref(ds.contigs), C.byref(ds.hits), ds.read_hit_off, ds.reads.n, C.byref(prm), C.byref(chain)) == 0
    assert L.orc_edge_support(C.byref(ds.contigs), C.byref(ds.hits), C.byref(prm), C.byref(chain), 0, ds.reads.n, C.byref(e)) == 0
    d = T.edges_to_dict(e, sides=False)
    assert m.shape[0] == len(d["key"]) > 0
    assert np.array_equal(m[:, 0], d["key"]) and np.array_equal(m[:, 1], d["lr"]) and np.array_equal(m[:, 2], d["cmp_head"]) and np.array_equal(m[:, 3], d["cmp_tail"])


def test_shard_bounds_cover_all_reads(built):
    import ctypes as C
    from haslr_amd import distributed as hd
    rho = np.array([0, 0, 5, 5, 9, 20, 20, 31], dtype=np.uint64)
    p = rho.ctypes.data_as(C.POINTER(C.c_uint64))
    for world in (1, 2, 3, 4, 8):
        b = hd.shard_bounds(p, 7, world)
        assert b[0] == 0 and b[-1] == 7 and len(b) == world + 1 and all(x <= y for x, y in zip(b, b[1:]))


FULL_WORKER = r'''
import ctypes as C, hashlib, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from haslr_amd import host, distributed as hd, ctypes_defs as T
import orclib
pre, out = sys.argv[2], sys.argv[3]
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
prm = ds.params()
ob = orclib.OracleBackend(ds, 4)
b = hd.shard_bounds(ds.read_hit_off, ds.reads.n, world)
ob.set_read_shard(b[rank], b[rank + 1])          # chain_reads hands the pipeline this rank's reads only, like a HIP context with a read shard


class OracleRecords:
    """the record source ShardedBackend drives (emit / export / import_), computed by the oracle: 32-byte records"""
    rec_bytes = 32

    def emit(self):
        d = ob.shard_edges(prm, b[rank], b[rank + 1])
        lr, tw = d["lr"] & 0x7fffffff, d["lr"] >> 31
        pos = np.where(tw == 0, d["cmp_head"], d["cmp_tail"])
        order = np.lexsort((tw, pos, lr))         # emission order: read asc, pair asc, forward before twin
        self.rec = np.zeros((len(order), 4), dtype=np.uint64)
        for k, name in enumerate(("key", "lr", "cmp_head", "cmp_tail")):
            self.rec[:, k] = d[name][order]
        return len(order)

    def export(self, n):
        return torch.from_numpy(self.rec.view(np.uint8).reshape(-1).copy()) if n else torch.zeros(1, dtype=torch.uint8)

    def import_(self, merged, total, out):
        m = merged.numpy().view(np.uint64).reshape(-1, 4)
        assert m.shape[0] == total
        m = m[np.argsort(m[:, 0], kind="stable")]
        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(T.Params), C.POINTER(T.EdgesOut))
        rc = proto(ob.table.edge_support)(ob.table.ctx, C.byref(prm), out)     # the oracle's own multiset of ALL reads ...
        d = T.edges_to_dict(out.contents, sides=False)
        for k, name in enumerate(("key", "lr", "cmp_head", "cmp_tail")):       # ... which the gathered shards must reproduce, order included
            assert np.array_equal(m[:, k], d[name].astype(np.uint64)), name
        return rc


be = hd.ShardedBackend(ob.table, OracleRecords())
dev = torch.device("cpu")
run = hd.run_sharded(ds, prm, be, b[rank], rank, world, dev, out_dir=out if rank == 0 else None, assemble=False)
assert be.error is None
loose = os.environ.get("HASLR_TEST_IDLE_RANKS") is not None     # more ranks than edges: some ranks have nothing to compute
assert run.results_missing == 0 and run.n_edges_total > (0 if loose else 4)
if not loose:
    assert 0 < run.n_edges < run.n_edges_total, (run.n_edges, run.n_edges_total)   # every rank worked on a part of the queue only
counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
dist.all_gather(counts, torch.tensor([run.n_edges], dtype=torch.int64))
assert sum(int(c) for c in counts) == run.n_edges_total               # the shares partition the work queue
if loose:
    assert min(int(c) for c in counts) == 0, [int(c) for c in counts]
run.assemble()
fa = run.assembly_fasta()
# a run that has not been given the other ranks' results must refuse to stitch
lone = host.Run(ds, prm, be.table, None)
lone.set_edge_shard(rank, world); lone.set_read_shard(b[rank])
lone.chain(); lone.graph(); lone.coords(); lone.consensus()
assert lone.results_missing == run.n_edges_total - run.n_edges
if lone.results_missing:
    try:
        lone.assemble()
        raise SystemExit("assemble accepted an incomplete result set")
    except host.HostError as e:
        assert "have no coordinates" in str(e), str(e)
# all ranks stitched the same assembly
h = torch.frombuffer(bytearray(hashlib.sha256(fa.encode()).digest()), dtype=torch.uint8)
hs = [torch.zeros(32, dtype=torch.uint8) for _ in range(world)]
dist.all_gather(hs, h)
assert all(torch.equal(hs[0], x) for x in hs)
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_pipeline_writes_the_single_rank_outputs(sim, built, tmp_path):
    import orclib
    import util
    from haslr_amd import host
    pre = sim("--genome-len", "150000", "--seed", "41", "--variant-per-mb", "60", "--cov", "9")
    w = tmp_path / "worker_full.py"
    w.write_text(FULL_WORKER)
    out2 = str(tmp_path / "two_ranks")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29519", str(w), ROOT, pre, out2], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
    ob = orclib.OracleBackend(ds, 4)
    out1 = str(tmp_path / "one_rank")
    run = host.Run(ds, ds.params(), ob.table, out1)
    run.all()
    names = sorted(os.listdir(out1))
    assert "asm.final.fa" in names and "asm.final.ann" in names and "compact_uniq.txt" in names and sum(n.endswith(".gfa") for n in names) == 6
    assert sorted(os.listdir(out2)) == names
    assert util.compare_dirs(out1, out2) == []
    assert len(run.assembly_fasta()) > 50000 and run.n_edges > 8
    run.close(); ob.close(); ds.close()


def test_four_ranks_some_without_edges(sim, built, tmp_path):
    """A data set with fewer surviving edges than ranks: the idle ranks contribute empty result blobs (and possibly no
    records), every rank still ends with the single-rank assembly."""
    import orclib
    import util
    from haslr_amd import host
    pre = sim("--genome-len", "26000", "--seed", "12", "--cov", "7", "--gap-median", "400")
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
    ob = orclib.OracleBackend(ds, 2)
    out1 = str(tmp_path / "one_rank")
    run = host.Run(ds, ds.params(), ob.table, out1)
    run.all()
    assert 0 < run.n_edges < 4, run.n_edges        # (the point of the case; pick another seed if the simulator changes)
    w = tmp_path / "worker_full.py"
    w.write_text(FULL_WORKER)
    out4 = str(tmp_path / "four_ranks")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HASLR_TEST_IDLE_RANKS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1",
                        "--master-port", "29523", str(w), ROOT, pre, out4], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert util.compare_dirs(out1, out4) == []
    run.close(); ob.close(); ds.close()


FAIL_WORKER = r'''
import ctypes as C, os, sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from haslr_amd import host, distributed as hd, ctypes_defs as T
import orclib
pre = sys.argv[2]
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
prm = ds.params()
ob = orclib.OracleBackend(ds, 2)
b = hd.shard_bounds(ds.read_hit_off, ds.reads.n, world)
ob.set_read_shard(b[rank], b[rank + 1])
table = T.Backend()
C.memmove(C.byref(table), C.byref(ob.table), C.sizeof(T.Backend))
proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(T.CoordsOut))
cb = proto(lambda *_: -1)
fail_at = sys.argv[3]
if rank == 1 and fail_at == "coords":
    table.edge_coords = C.cast(cb, C.c_void_p).value      # this rank's coordinate operator fails
if rank == 1 and fail_at == "chain":
    table.chain_reads = C.cast(cb, C.c_void_p).value      # this rank's chain operator fails: its verdict travels with the count round of the record exchange
if rank == 1 and fail_at == "graph-early":                # the graph stage fails BEFORE it reaches the record exchange (the others are in that count round, not the results one)
    def early(self):
        raise host.HostError("graph stage failed before the record exchange")
    host.Run.graph = early


class Rec:                                              # the oracle's edge_support already returns the merged multiset: nothing to exchange
    rec_bytes = 32
    def emit(self): return 0
    def export(self, n): return torch.zeros(1, dtype=torch.uint8)
    def import_(self, merged, total, out):
        if rank == 1 and fail_at == "graph":             # AFTER the record all-gather (the import, the graph build, rank 0's files): this rank alone fails
            return -1
        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(T.Params), C.POINTER(T.EdgesOut))
        return proto(ob.table.edge_support)(ob.table.ctx, C.byref(prm), out)


be = hd.ShardedBackend(table, Rec())
t0 = time.time()
try:
    hd.run_sharded(ds, prm, be, b[rank], rank, world, torch.device("cpu"), out_dir=None, assemble=False)
    print(f"rank {rank}: NO ERROR", flush=True)
    code = 0
except Exception as e:
    print(f"rank {rank}: stopped after {time.time() - t0:.1f} s: {e}", flush=True)
    code = 7
os._exit(code)
'''


@pytest.mark.parametrize("fail_at,port", [("coords", "29527"), ("graph", "29529"), ("chain", "29531"), ("graph-early", "29533")])
def test_failure_on_one_rank_stops_every_rank(sim, built, tmp_path, fail_at, port):
    """a rank whose chain stage, whose coordinate stage - or whose graph stage, after the record all-gather - fails: the count round ahead of
    each of the two all-gathers carries every rank's verdict on what it did since the previous one, so every rank raises within seconds
    instead of waiting in the next collective for the gloo / RCCL watchdog"""
    pre = sim("--genome-len", "120000", "--seed", "77", "--variant-per-mb", "30", "--cov", "12")
    w = tmp_path / "worker_fail.py"
    w.write_text(FAIL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
                        "--master-port", port, str(w), ROOT, pre, fail_at], env=env, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode != 0
    for k in range(3):
        assert f"rank {k}: stopped after" in out, out[-3000:]
    assert "NO ERROR" not in out
    assert "rank 0: stopped" in out and "another rank" in out
