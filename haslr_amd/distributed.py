"""Multi-GPU orchestration: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The path shards twice with ONE exchange in between:
  phase 1  long reads are split into contiguous id ranges balanced on raw-hit count; every rank filters /
           sorts / trims / chains its reads and emits their edge-support records
  exchange one all-gather of the packed records (ranks own ascending read ranges, so concatenation in rank
           order followed by the stable key sort reproduces the reference's per-edge support order)
  phase 2  every rank sorts + segments the full multiset and cleans the (small) graph redundantly, then
           computes coordinates and POA consensus for its share of the surviving edges.
Raw inputs (CIGAR ops, packed reads) are replicated on every GPU, so records only carry indices.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import ctypes_defs as T


def shard_bounds(read_hit_off, n_reads, world):
    """Contiguous read-id ranges with ~equal numbers of raw PAF records. Returns world+1 boundaries."""
    rho = np.ctypeslib.as_array(read_hit_off, shape=(n_reads + 1,))
    total = int(rho[-1])
    bounds = [0]
    for r in range(1, world):
        bounds.append(int(np.searchsorted(rho, total * r // world, side="left")))
    bounds.append(n_reads)
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds


def allgather_records(local: torch.Tensor, n_local: int, rec_bytes: int, group=None):
    """All-gather of variable-length packed record buffers (uint8 tensors of n_local*rec_bytes bytes).
    Returns (merged tensor in rank order, total record count). One data collective (+ a count exchange)."""
    world = dist.get_world_size(group)
    dev = local.device
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([n_local], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, mine, group=group) if dev.type == "cuda" else dist.all_gather(list(counts.split(1)), mine, group=group)
    counts_h = counts.cpu().tolist()
    cap = max(max(counts_h), 1) * rec_bytes
    padded = torch.zeros(cap, dtype=torch.uint8, device=dev)
    padded[: n_local * rec_bytes] = local[: n_local * rec_bytes]
    gathered = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(gathered, padded, group=group)
    merged = torch.cat([g[: c * rec_bytes] for g, c in zip(gathered, counts_h)]) if sum(counts_h) else torch.zeros(0, dtype=torch.uint8, device=dev)
    return merged.contiguous(), int(sum(counts_h))


class ShardedBackend:
    """Backend table for the host pipeline in a multi-GPU run: chain/coords/POA are the context's own
    operators (restricted to the read shard set on the context); edge_support = emit + all-gather + import."""

    def __init__(self, ctx, params, group=None):
        from . import hip
        self.ctx, self.params, self.group = ctx, params, group
        self.rec_bytes = hip.records_bytes()
        self.table = T.Backend()
        C.memmove(C.byref(self.table), C.byref(ctx.table), C.sizeof(T.Backend))
        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(T.Params), C.POINTER(T.EdgesOut))
        self._cb = proto(self._edge_support)
        self.table.edge_support = C.cast(self._cb, C.c_void_p).value
        self.exchange_bytes = 0

    def _edge_support(self, _ctx, prm, out):
        from . import hip
        try:
            n = self.ctx.edge_emit(self.params)
            dev = torch.device("cuda", torch.cuda.current_device())
            local = torch.empty(max(n, 1) * self.rec_bytes, dtype=torch.uint8, device=dev)
            self.ctx.edge_records_export(C.c_void_p(local.data_ptr()), n)
            merged, total = allgather_records(local, n, self.rec_bytes, self.group)
            self.exchange_bytes = total * self.rec_bytes
            torch.cuda.synchronize()
            rc = hip.lib().hx_edge_records_import(self.ctx._h, C.c_void_p(merged.data_ptr()), total, out)
            return rc
        except Exception as e:  # noqa: BLE001 - must not propagate through the C callback
            print(f"[ERROR] sharded edge_support: {e}", flush=True)
            return -1
