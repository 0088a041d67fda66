"""world_size-2 CPU test (gloo) of the multi-GPU exchange logic: read sharding + all-gather of packed
edge-support records in rank order + stable key sort == the unsharded multiset. The per-shard compute is done
by the CPU oracle here (no GPU in this container); the collective code path is haslr_amd.distributed's."""
import os
import subprocess
import sys

import numpy as np

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
